"""A small dataflow check over gfx950 assembly (test infrastructure): every register a vector-memory load writes must be covered by an
s_waitcnt vmcnt before an instruction reads it, on EVERY path — and every LDS-DMA (global_load_lds_*) before the next s_barrier.

vmcnt counts a wave's vector-memory operations (loads, stores, LDS-DMA, scratch) in issue order; `s_waitcnt vmcnt(N)` waits until at most
the N youngest are outstanding.  State per pending load destination: a lower bound p on the number of vector-memory operations issued
after it (min over paths); vmcnt(N) retires it iff p >= N.  Joins take the union of the pending sets with the minimum p, iterated to a
fixed point over the control-flow graph of the function's basic blocks."""
import re

VMEM_LOAD = re.compile(r"^(buffer_load|global_load|flat_load|scratch_load)_(?!lds)")
VMEM_DMA = re.compile(r"^(global_load_lds|buffer_load_\w+\s.*\blds\b)")
VMEM_OTHER = re.compile(r"^(buffer_store|global_store|flat_store|scratch_store|global_atomic|buffer_atomic|flat_atomic)")
CAP = 64


def _regs(tok):
    m = re.fullmatch(r"v(\d+)", tok)
    if m:
        return [int(m.group(1))]
    m = re.fullmatch(r"v\[(\d+):(\d+)\]", tok)
    if m:
        return list(range(int(m.group(1)), int(m.group(2)) + 1))
    return []


def parse_function(lines):
    """-> (blocks: list of (label, [instr]), succ: {index: [indices]}); instr = (opcode, [operand tokens], raw)"""
    blocks, cur, label = [], [], "entry"
    for raw in lines:
        line = raw.split(";")[0].strip()
        if not line:
            continue
        m = re.match(r"^(\.LBB\w+):", line)
        if m:
            blocks.append((label, cur))
            label, cur = m.group(1), []
            continue
        if line.startswith(".") or line.endswith(":"):
            continue
        parts = line.split(None, 1)
        ops = [t.strip() for t in parts[1].split(",")] if len(parts) > 1 else []
        ops = [t.split()[0] if t else t for t in ops]   # drop modifiers such as "offen offset:16"
        cur.append((parts[0], ops, line))
    blocks.append((label, cur))
    index = {lab: i for i, (lab, _) in enumerate(blocks)}
    succ = {}
    for i, (lab, ins) in enumerate(blocks):
        out = []
        fall = True
        for op, ops, _ in ins:
            if op == "s_branch":
                out.append(index[ops[0]])
                fall = False
            elif op.startswith("s_cbranch"):
                out.append(index[ops[0]])
            elif op == "s_endpgm":
                fall = False
        if fall and i + 1 < len(blocks):
            out.append(i + 1)
        succ[i] = out
    return blocks, succ


def _step(state, op, ops, raw, violations, where):
    # reads
    first_is_src = bool(VMEM_OTHER.match(op)) or op.startswith("ds_write") or op.startswith("v_cmp") or bool(VMEM_DMA.match(raw))
    srcs = ops if first_is_src else ops[1:]
    for t in srcs:
        for r in _regs(t):
            if r in state:
                violations.append((where, raw, f"v{r} read while its load may be outstanding (p >= {state[r]})"))
    if op == "s_barrier" and "dma" in state:
        violations.append((where, raw, "s_barrier with an LDS-DMA possibly outstanding"))
    if op == "s_waitcnt":
        m = re.search(r"vmcnt\((\d+)\)", raw)
        n = int(m.group(1)) if m else (0 if re.fullmatch(r"s_waitcnt\s+(0|0x0+)", raw) else None)
        if n is not None:
            for k in [k for k, p in state.items() if p >= n]:
                del state[k]
        return
    is_load, is_dma, is_other = bool(VMEM_LOAD.match(op)), bool(VMEM_DMA.match(raw)), bool(VMEM_OTHER.match(op))
    if is_load or is_dma or is_other:
        for k in state:
            state[k] = min(state[k] + 1, CAP)
        if is_dma:
            state["dma"] = 0
        elif is_load and ops:
            for r in _regs(ops[0]):
                state[r] = 0
    elif ops and not first_is_src:   # a plain write to a register ends its pending state only if the hardware interlocks; ignore
        pass


def check(lines):
    blocks, succ = parse_function(lines)
    n = len(blocks)
    ins = [None] * n
    ins[0] = {}
    work = [0]
    outs = [None] * n
    while work:
        b = work.pop()
        st = dict(ins[b])
        sink = []
        for op, ops, raw in blocks[b][1]:
            _step(st, op, ops, raw, sink, blocks[b][0])
        if outs[b] == st:
            continue
        outs[b] = st
        for s in succ[b]:
            if ins[s] is None:
                ins[s] = dict(st)
                work.append(s)
            else:
                merged = dict(ins[s])
                changed = False
                for k, p in st.items():
                    if k not in merged or p < merged[k]:
                        merged[k] = p
                        changed = True
                if changed:
                    ins[s] = merged
                    work.append(s)
    violations = []
    for b in range(n):
        if ins[b] is None:
            continue
        st = dict(ins[b])
        for op, ops, raw in blocks[b][1]:
            _step(st, op, ops, raw, violations, blocks[b][0])
    return violations
