#!/usr/bin/env python3
"""tools/isa_histogram.py -- dynamic instruction histogram of one gfx950 kernel from hipcc's assembly.

The dominant kernels of the proving path are integer-VALU bound, so their roofline is an ISSUE ceiling: the
instruction mix of the kernel (what this tool counts) x the measured issue cost of each opcode
(profiles/r02_ubench_issue_cost.json, written by tools/ubench_issue.hip on the GPU) = the fewest cycles a SIMD can
spend on one wave of the kernel.  bench.py divides that by the measured cycles per wave.

Method: compile the .hip file with `hipcc --cuda-device-only -S`, cut out the kernel, split it into basic blocks,
read the loop nest from the asm printer's own annotations, weight every block by the product of the trip counts of the loops
that contain it (trip counts are given on the command line -- they are compile-time constants of the algorithm, e.g.
the 3 / 7 / 3 round loops of poseidon_permute and the 33 absorb steps of a 262-column row), and count opcodes.
The weighted VALU total is cross-checked by bench.py against rocprofv3's SQ_INSTS_VALU / SQ_WAVES of the same kernel.

Usage:
  python tools/isa_histogram.py zkm_amd/csrc/hash.hip k_merkle_leaves --trips 33,3,7,3,3,7,3,1 --only-loop 0 \
      --out profiles/r02_isa_merkle_leaves.json
`--trips` lists one trip count per loop in order of first appearance (run without it to see the loops);
`--only-loop K` restricts the histogram to ONE iteration of the K-th loop (the steady state of the kernel).
"""
import argparse
import json
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def compile_asm(src, arch="gfx950"):
    out = tempfile.NamedTemporaryFile(suffix=".s", delete=False).name
    cmd = ["hipcc", "-O3", "-std=c++17", "--offload-arch=" + arch, "-Wno-unused-function", "-mllvm", "-amdgpu-mfma-vgpr-form=1", "--cuda-device-only", "-S", "-o", out, src]   # (the flags of csrc/Makefile)
    subprocess.check_call(cmd, stderr=subprocess.DEVNULL)
    text = open(out).read()
    os.unlink(out)
    return text


def cut_kernel(text, name):
    """Lines of the first kernel whose mangled symbol contains `name`, from its label to s_endpgm (inclusive)."""
    lines = text.split("\n")
    start = None
    for i, l in enumerate(lines):
        m = re.match(r"^(_Z\w*%s\w*):" % re.escape(name), l)
        if m:
            start, sym = i, m.group(1)
            break
    if start is None:
        raise SystemExit("kernel %s not found" % name)
    end = start
    for j in range(start, len(lines)):
        if lines[j].strip().startswith(".end_amdhsa_kernel") or lines[j].startswith(".Lfunc_end"):
            break
        end = j
    return sym, lines[start + 1:end + 1]


def parse(lines):
    """-> (blocks, parent): basic blocks in layout order, each {"label", "ops", "loop"} where "loop" is the label of the
    innermost loop header the block belongs to (from the asm printer's own loop annotations, which survive loop rotation and
    out-of-line blocks), and parent[header] = enclosing loop's header or None."""
    blocks, parent = [], {}
    cur = {"label": None, "ops": [], "loop": None, "name": None}

    def annotate(block, note, own_label):
        m = re.search(r"in Loop: Header=(BB\w+)", note)
        if m:
            block["loop"] = ".L" + m.group(1)
        if "Loop Header" in note and own_label:
            block["loop"] = own_label
            ps = re.findall(r"Parent Loop (BB\w+) Depth=(\d+)", note)
            parent[own_label] = ".L" + max(ps, key=lambda q: int(q[1]))[0] if ps else None

    for li, l in enumerate(lines):
        s = l.split(";")[0].strip()
        lab = re.match(r"^(\.LBB\w+):", s)
        bb = re.match(r"^\s*;\s*%bb\.\d+:", l)
        if lab or bb:
            if cur["ops"] or cur["label"]:
                blocks.append(cur)
            note, k = l, li + 1
            while k < len(lines) and lines[k].lstrip().startswith(";") and not re.match(r"^\s*;\s*%bb\.\d+:", lines[k]):
                note += lines[k]
                k += 1
            cur = {"label": lab.group(1) if lab else None, "ops": [], "loop": None,
                   "name": lab.group(1) if lab else re.search(r"%bb\.\d+", l).group(0)}
            annotate(cur, note, cur["label"])
            continue
        mreg = re.search(r"ZKM_REGION (\w+)", l)
        if mreg:
            # a region marker splits the block: what follows (in layout order) belongs to the named region
            if cur["ops"]:
                blocks.append(cur)
                cur = {"label": None, "ops": [], "loop": cur["loop"], "name": cur.get("name"), "cold": cur.get("cold", False)}
            cur["region"] = mreg.group(1)
        if "ZKM_COLD" in l:
            cur["cold"] = True   # source-level marker (gl_dev.h): a block behind a never-taken branch
        if not s or s.startswith("."):
            continue
        op = s.split()[0]
        cur["ops"].append(op)
        if op.startswith("s_cbranch") or op == "s_branch":
            blocks.append(cur)
            # a fall-through block without its own annotation stays in the loop of the block it follows
            cur = {"label": None, "ops": [], "loop": cur["loop"], "name": cur.get("name")}
    if cur["ops"] or cur["label"]:
        blocks.append(cur)
    return blocks, parent


def find_loops(blocks, parent):
    """Loop headers in order of first appearance (layout order of their first member block)."""
    seen = []
    for b in blocks:
        if b["loop"] and b["loop"] not in seen:
            seen.append(b["loop"])
    for h in list(seen):
        q = parent.get(h)
        while q and q not in seen:
            seen.append(q)
            q = parent.get(q)
    return seen


def classify(op):
    if op.startswith("v_"):
        return "valu"
    if op.startswith("s_"):
        return "salu"
    if op.startswith(("global_", "buffer_", "flat_", "scratch_")):
        return "vmem"
    if op.startswith("ds_"):
        return "lds"
    return "other"


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("source")
    ap.add_argument("kernel")
    ap.add_argument("--trips", default="", help="comma-separated trip counts, one per loop in header order")
    ap.add_argument("--only-loop", type=int, default=None, help="restrict to the body of the K-th loop (header order)")
    ap.add_argument("--exec", default="", help="NAME=count,...: executions of a block (label or %%bb.N; a branch-split block keeps its name) "
                    "per iteration of the --only-loop loop, for code under a condition the trip counts cannot express")
    ap.add_argument("--region-exec", default="", help="REGION=count,...: executions per iteration of the --only-loop loop of the code that "
                    "follows a `; ZKM_REGION name` marker (source-level markers, e.g. poseidon_dev.h) up to the next marker, in layout order")
    ap.add_argument("--cost", default=os.path.join(ROOT, "profiles", "r02_ubench_issue_cost.json"))
    ap.add_argument("--out", default=None)
    args = ap.parse_args()

    sym, lines = cut_kernel(compile_asm(os.path.join(ROOT, args.source) if not os.path.isabs(args.source) else args.source), args.kernel)
    blocks, parent = parse(lines)
    loops = find_loops(blocks, parent)
    trips = [int(x) for x in args.trips.split(",") if x]
    if trips and len(trips) != len(loops):
        print("loops found (header, parent):", [(h, parent.get(h)) for h in loops], file=sys.stderr)
        raise SystemExit("--trips needs %d entries" % len(loops))
    trip = dict(zip(loops, trips or [1] * len(loops)))

    def chain(h):
        out = []
        while h:
            out.append(h)
            h = parent.get(h)
        return out
    weight = []
    for b in blocks:
        w = 1
        for h in chain(b["loop"]):
            w *= trip[h]
        weight.append(w)
    keep = [True] * len(blocks)
    if args.only_loop is not None:
        root = loops[args.only_loop]
        keep = [root in chain(b["loop"]) for b in blocks]
        weight = [w // trip[root] if k else w for w, k in zip(weight, keep)]   # per iteration of the selected loop
    for i, b in enumerate(blocks):   # cold blocks, and the fall-through pieces a branch splits them into, do not execute
        if b.get("cold"):
            weight[i] = 0
    rexec = dict((kv.split("=")[0], float(kv.split("=")[1])) for kv in args.region_exec.split(",") if kv)
    region = None
    for i, b in enumerate(blocks):
        region = b.get("region", region)
        if rexec and keep[i] and not b.get("cold"):
            if region in rexec:
                weight[i] = rexec[region]
    overrides = dict((kv.split("=")[0], float(kv.split("=")[1])) for kv in args.exec.split(",") if kv)
    for i, b in enumerate(blocks):
        if b.get("name") in overrides:
            weight[i] = overrides[b["name"]]
    hist = {}
    for i in range(len(blocks)):
        if not keep[i]:
            continue
        for op in blocks[i]["ops"]:
            hist[op] = hist.get(op, 0) + weight[i]
    classes = {}
    for op, n in hist.items():
        classes[classify(op)] = classes.get(classify(op), 0) + n
    out = {"source": args.source, "kernel": sym, "loops": [{"header": h, "parent": parent.get(h), "trip": trip[h]} for h in loops],
           "region": "one iteration of loop %s" % loops[args.only_loop] if args.only_loop is not None else "whole kernel",
           "block_executions_override": overrides, "class_totals": classes, "histogram": dict(sorted(hist.items(), key=lambda kv: -kv[1]))}
    if os.path.exists(args.cost):
        cost = json.load(open(args.cost))["cycles_per_wave_instr"]
        default = cost.get("_default_vop3", 4.4)
        cyc, missing = 0.0, {}
        for op, n in hist.items():
            if classify(op) != "valu":
                continue
            base = re.sub(r"_e(32|64)$", "", op)
            c = cost.get(base)
            if c is None:
                missing[base] = missing.get(base, 0) + n
                c = default
            cyc += c * n
        out["valu_issue_cycles_per_wave"] = cyc
        out["valu_instr_per_wave"] = classes.get("valu", 0)
        out["mean_issue_cost"] = cyc / max(1, classes.get("valu", 0))
        out["opcodes_without_measured_cost"] = missing
        out["cost_source"] = os.path.relpath(args.cost, ROOT)
    js = json.dumps(out, indent=1)
    if args.out:
        open(args.out, "w").write(js + "\n")
    print(js if not args.out else "wrote %s: %d VALU instr/wave in region" % (args.out, classes.get("valu", 0)))


if __name__ == "__main__":
    main()
