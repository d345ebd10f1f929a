"""Static instruction counts of render_kernel_8<4u> per phase of its main loop (development aid, runs
where hipcc is: no GPU needed).  Comment markers are put at the phase boundaries of a copy of
kernels.hip, the copy is compiled to assembly, and the instructions between markers are counted by
issue class (the classes of tools/microbench: fast ~2.3 cycles, normal ~4.1, transcendental ~8.2,
64-bit multiply ~5).  Markers are `asm volatile` comments: the compiler keeps them in order but may
move a few independent vector instructions across them.
  python tools/static_blocks.py > profiles/r03_static_blocks.txt
  python tools/static_blocks.py --dump NODE_STEPS_x2      (the instructions of one phase)"""
import collections, os, re, subprocess, sys, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src = open(os.path.join(ROOT, "raytracing_amd/csrc/kernels.hip")).read()
MARKS = [
    ("    // ---- 1. shade the finished rays", "SHADE"),
    ("    // ---- 2. dead lanes claim the next", "CLAIM"),
    ("    // ---- 3. one random event for every lane", "PHILOX"),
    ("      bool go = true;\n", "TRIG+CAMERA"),
    ("      } else {\n        V3<R> dir;\n        go = scatter_direction", "SCATTER"),
    ("      if (go) {\n        bounce++;", "RAY_SETUP+RAY_START"),
    ("    if (__ballot(state != ST_DONE) == 0ull) break;  // the queue is empty", "TRAVERSAL_HEAD"),
    ("      for (;;) {\n        const bool inner = tv.cur < RF::LEAF;", "NODE_STEPS_x2"),
    ("      if (ml != 0ull) {\n        if (leaf) {", "LEAF_STEP_PAIR"),
    ("      if (__popcll(__ballot(tv.cur != RF::EMPTY)) <= stop) break;", "TRAVERSAL_POLICY"),
]
for anchor, name in MARKS:
    assert anchor in src, anchor
    src = src.replace(anchor, '    asm volatile("; MARK_%s" ::: "memory");\n' % name + anchor, 1)
src = src.replace('"../../include/hiptrace.h"', '"%s/include/hiptrace.h"' % ROOT)
with tempfile.TemporaryDirectory() as tmp:
    f = os.path.join(tmp, "k.hip")
    open(f, "w").write(src)
    subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-fno-slp-vectorize",
                           "--offload-device-only", "-S", "-I", os.path.join(ROOT, "raytracing_amd/csrc"), "-o", f + ".s", f],
                          stderr=subprocess.DEVNULL)
    text = open(f + ".s").read()
start = text.index("_ZN2ht15render_kernel_8ILj4EEEvNS_10RenderArgsENS_6CamDevIfEE:")
lines = text[start:text.index(".Lfunc_end", start)].split("\n")
marks = [(0, "PROLOGUE")] + [(i, l.split("MARK_")[1].strip()) for i, l in enumerate(lines) if "MARK_" in l] + [(len(lines), "")]
FAST = {"v_add_f32", "v_sub_f32", "v_mul_f32", "v_fmac_f32", "v_fma_f32", "v_mov_b32", "v_and_b32", "v_or_b32", "v_xor_b32",
        "v_add_u32", "v_sub_u32", "v_lshrrev_b32", "v_subrev_f32", "v_subrev_u32", "v_not_b32", "v_xor3_b32"}
TRANS = {"v_rcp_f32", "v_sqrt_f32", "v_rsq_f32", "v_sin_f32", "v_cos_f32", "v_exp_f32", "v_log_f32", "v_rcp_iflag_f32"}
if len(sys.argv) > 2 and sys.argv[1] == "--dump":  # the instructions of one phase
    for (a, name), (b, _) in zip(marks, marks[1:]):
        if name == sys.argv[2]:
            print("\n".join(l for l in lines[a:b] if l.strip() and not l.strip().startswith((";", "."))))
    sys.exit(0)
print("render_kernel_8<4u>, static instructions per phase (VALU by issue class, estimated issue cycles of one pass)")
print(f"{'phase':22s} {'VALU':>5s} {'fast':>5s} {'norm':>5s} {'trans':>5s} {'mad64':>5s} {'lane':>5s} {'~cyc':>6s} | {'SALU':>5s} {'brch':>5s} {'LDS':>4s} {'VMEM':>4s}")
for (a, name), (b, _) in zip(marks, marks[1:]):
    c = collections.Counter()
    for l in lines[a:b]:
        m = re.match(r"\s+([a-z][a-z0-9_]+)", l)
        if not m:
            continue
        op = re.sub(r"_(e32|e64|sdwa|dpp)$", "", m.group(1))
        if op.startswith("v_"):
            k = ("trans" if op in TRANS else "fast" if op in FAST else "lane" if op in ("v_readlane_b32", "v_writelane_b32", "v_readfirstlane_b32")
                 else "mad64" if op == "v_mad_u64_u32" else "norm")
            c[k] += 1
        elif op.startswith("s_"):
            c["wait" if op in ("s_waitcnt", "s_nop") else "brch" if "branch" in op else "SALU"] += 1
        elif op.startswith("ds_"):
            c["LDS"] += 1
        elif op.split("_")[0] in ("global", "flat", "buffer"):
            c["VMEM"] += 1
    valu = c["fast"] + c["norm"] + c["trans"] + c["mad64"] + c["lane"]
    cyc = c["fast"] * 2.3 + c["norm"] * 4.1 + c["trans"] * 8.2 + c["mad64"] * 5 + c["lane"] * 4.1
    print(f"{name:22s} {valu:5d} {c['fast']:5d} {c['norm']:5d} {c['trans']:5d} {c['mad64']:5d} {c['lane']:5d} {cyc:6.0f} | {c['SALU']:5d} {c['brch']:5d} {c['LDS']:4d} {c['VMEM']:4d}")
