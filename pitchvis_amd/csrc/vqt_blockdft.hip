// vqt_blockdft.hip — hop-block DFT path (fp32 MFMA GEMM + phase combine).  Placeholder until the
// kernels land: reports "not applicable" so PVQ_ALGO_AUTO always takes the LDS-FFT path.
#include "vqt_engine.hpp"
#include "device_tables.hpp"

namespace pvq {

struct BlockDftTables {};

void free_blockdft_tables(BlockDftTables* t) { delete t; }

bool Vqt::blockdft_applicable(size_t) const { return false; }

pvq_status Vqt::prepare_blockdft(size_t) { return PVQ_ERR_UNSUPPORTED; }

pvq_status Vqt::launch_blockdft_path(const float*, size_t, size_t, size_t, float*, float*, hipStream_t) {
    set_last_error("block-DFT path not built");
    return PVQ_ERR_UNSUPPORTED;
}

}  // namespace pvq
