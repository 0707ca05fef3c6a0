"""Occupancy guard for the hand-written kernels (no GPU needed: hipcc cross-compiles and reports register use).
The fused GEMM + tree kernel runs two 512-thread workgroups per CU = four waves per SIMD, which holds only up to 128
vector registers per lane; one register more silently halves its occupancy (measured: 0.36 -> 0.42 ms per launch)."""
import os
import re
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HIPCC = "/opt/rocm/bin/hipcc"


@pytest.mark.skipif(not os.path.exists(HIPCC), reason="hipcc not installed")
def test_register_budgets_of_the_block_dft_kernels(tmp_path):
    src = os.path.join(ROOT, "pitchvis_amd", "csrc", "vqt_blockdft.hip")
    r = subprocess.run([HIPCC, "-O3", "-std=c++17", "-fPIC", "--offload-arch=gfx950", "-Rpass-analysis=kernel-resource-usage",
                        "-c", src, "-o", str(tmp_path / "x.o")], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    usage = {}
    name = None
    for line in r.stderr.splitlines():
        m = re.search(r"Function Name: (\S+)", line)
        if m:
            name = m.group(1)
            usage[name] = {}
        m = re.search(r"remark:\s+(VGPRs|AGPRs|ScratchSize \[bytes/lane\]|Occupancy \[waves/SIMD\]): (\d+)", line)
        if m and name:
            usage[name][m.group(1).split(" ")[0]] = int(m.group(2))

    def find(sub):
        hits = [v for k, v in usage.items() if sub in k]
        assert hits, (sub, list(usage))
        return hits

    for u in find("blockdft_gemm_treeILi256") + find("blockdft_gemm_treeILi128") + find("blockdft_gemm_tree_bf16x3ILi256"):
        # two 512-thread workgroups per CU; a dword or two of the tile set-up may spill, nothing between the matrix instructions may
        assert u["VGPRs"] + u.get("AGPRs", 0) <= 128 and u["ScratchSize"] <= 16, u
    asm = tmp_path / "x.s"
    r = subprocess.run([HIPCC, "-O3", "-std=c++17", "-fPIC", "--offload-arch=gfx950", "-S", "--cuda-device-only", src, "-o", str(asm)],
                       capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    body, inside = {}, None
    for line in asm.read_text().splitlines():
        m = re.match(r"(_ZN3pvq\w+):", line)
        if m:
            inside = m.group(1)
            body[inside] = []
        elif line.startswith(".Lfunc_end"):
            inside = None
        elif inside:
            body[inside].append(line)
    checked = 0
    for name, lines in body.items():
        if "blockdft_gemm_tree" not in name:
            continue
        mf = [i for i, l in enumerate(lines) if "v_mfma" in l]
        sc = [i for i, l in enumerate(lines) if re.match(r"\s+scratch_", l)]
        assert mf, name
        assert all(i < mf[0] or i > mf[-1] for i in sc), (name, "scratch access inside the K loop", [lines[i] for i in sc])
        checked += 1
    assert checked >= 3
    for u in find("blockdft_banddots8_dbILi8ELi4ELi260"):
        assert u["VGPRs"] + u.get("AGPRs", 0) <= 128 and u["ScratchSize"] == 0, u    # 8 waves x 2 workgroups per CU
