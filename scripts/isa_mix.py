"""Static instruction mix of one kernel from the compiler's assembly, per basic block.

    hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off --cuda-device-only -S \
        path_tracer_amd/csrc/ptgpu.hip -o /tmp/ptgpu.s
    python scripts/isa_mix.py /tmp/ptgpu.s k_traceILi1ELb0ELb1E [--blocks] [--weights FILE]

Classes: fp64 (v_*_f64 arithmetic incl. rcp/rsq/sqrt/div helpers), fp32 (v_*_f32, packed), int
(integer add/mul/shift/logic), sel (v_cndmask), cmp (v_cmp*), cvt, mov (v_mov, readlane ...),
lds (ds_*), vmem (global_/scratch_/flat_), salu (s_* arithmetic/logic/compare/mov), smem (s_load*),
branch (s_cbranch/s_branch), wait (s_waitcnt, s_nop, ...).  With --weights (JSON: block label ->
executions per wave, e.g. from the section counters) the block counts are weighted into a
per-wave dynamic mix; without it the whole-kernel static mix and the blocks that close a loop
(a backward branch) are printed."""
import collections
import json
import re
import sys

CLASSES = ["fp64", "fp32", "int", "sel", "cmp", "cvt", "mov", "lds", "vmem", "salu", "smem", "branch", "wait", "other"]


def classify(m):
    if m.startswith("v_"):
        if m.startswith("v_cndmask"):
            return "sel"
        if m.startswith("v_cmp") or m.startswith("v_cmpx"):
            return "cmp"
        if m.startswith("v_cvt"):
            return "cvt"
        if m.startswith(("v_mov", "v_readlane", "v_readfirstlane", "v_writelane", "v_accvgpr", "v_swap", "v_mbcnt", "v_bfrev")):
            return "mov"
        if "_f64" in m:
            return "fp64"
        if "_f32" in m or "_f16" in m or m.startswith("v_pk_"):
            return "fp32"
        return "int"
    if m.startswith("ds_"):
        return "lds"
    if m.startswith(("global_", "scratch_", "flat_", "buffer_")):
        return "vmem"
    if m.startswith("s_load") or m.startswith("s_buffer_load") or m.startswith("s_memtime") or m.startswith("s_memrealtime"):
        return "smem"
    if m.startswith(("s_cbranch", "s_branch", "s_endpgm", "s_setpc", "s_swappc")):
        return "branch"
    if m.startswith(("s_waitcnt", "s_nop", "s_barrier", "s_sleep", "s_setprio")):
        return "wait"
    if m.startswith("s_"):
        return "salu"
    return "other"


def parse(path, kernel):
    blocks = collections.OrderedDict()
    order = []
    cur = None
    inside = False
    with open(path) as f:
        for line in f:
            if not inside:
                if re.match(r"^_Z\S*%s\S*:" % re.escape(kernel), line):
                    inside = True
                    cur = "entry"
                    blocks[cur] = []
                    order.append(cur)
                continue
            if line.startswith("\t.end_amdhsa_kernel") or line.startswith(".Lfunc_end") or line.startswith("\t.section"):
                break
            m = re.match(r"^(\.LBB\d+_\d+):", line)
            if m:
                cur = m.group(1)
                blocks[cur] = []
                order.append(cur)
                continue
            m = re.match(r"^\t([a-z][a-z0-9_]+)\b(.*)", line)
            if m and not m.group(1).startswith("."):
                blocks[cur].append((m.group(1), m.group(2)))
    return blocks, order


def main():
    path, kernel = sys.argv[1], sys.argv[2]
    show_blocks = "--blocks" in sys.argv
    weights = None
    if "--weights" in sys.argv:
        with open(sys.argv[sys.argv.index("--weights") + 1]) as f:
            weights = json.load(f)
    blocks, order = parse(path, kernel)
    pos = {b: i for i, b in enumerate(order)}
    total = collections.Counter()
    dyn = collections.Counter()
    loops = []
    for b in order:
        c = collections.Counter(classify(m) for m, _ in blocks[b])
        total.update(c)
        if weights is not None:
            w = float(weights.get(b, 0.0))
            for k, v in c.items():
                dyn[k] += w * v
        for m, ops in blocks[b]:
            if m.startswith(("s_cbranch", "s_branch")):
                t = re.search(r"(\.LBB\d+_\d+)", ops)
                if t and t.group(1) in pos and pos[t.group(1)] <= pos[b]:
                    loops.append((t.group(1), b))
        if show_blocks:
            print(b, len(blocks[b]), dict(c))
    n = sum(total.values())
    print(f"kernel {kernel}: {n} instructions in {len(order)} blocks (static)")
    valu = sum(total[k] for k in ("fp64", "fp32", "int", "sel", "cmp", "cvt", "mov"))
    for k in CLASSES:
        if total[k]:
            print(f"  {k:7s} {total[k]:6d}  {100.0 * total[k] / n:5.1f} % of all" + (f"  {100.0 * total[k] / valu:5.1f} % of VALU" if k in ("fp64", "fp32", "int", "sel", "cmp", "cvt", "mov") else ""))
    print("loops (head .. closing block): static instructions inside, by class")
    for head, tail in loops:
        c = collections.Counter()
        for b in order[pos[head]: pos[tail] + 1]:
            c.update(classify(m) for m, _ in blocks[b])
        print(f"  {head} .. {tail}: {sum(c.values())} instr  " + " ".join(f"{k}={c[k]}" for k in CLASSES if c[k]))
    if weights is not None:
        n = sum(dyn.values())
        print(f"dynamic mix per wave ({n:.0f} instructions):")
        for k in CLASSES:
            if dyn[k]:
                print(f"  {k:7s} {dyn[k]:9.0f}  {100.0 * dyn[k] / n:5.1f} %")


if __name__ == "__main__":
    main()
