#!/usr/bin/env python3
"""Static VALU count of the 6-DoF step kernel's RK4 sub-step loop from `hipcc -S` output (no GPU needed).

    python tools/isa_count.py [--zoh] [--noflow] [--fixed] [-D...]      compiles csrc/mvrl_rov6.hip for gfx950 and prints, for the baked
                                                            FAITHFUL+flow instance: VALU instructions in the sub-step loop on the
                                                            fast path, in the divergent blocks (full sincos of single lanes), and
                                                            outside the loop; registers, scratch.
The executed count (SQ_INSTS_VALU / SQ_WAVES on the GPU) is about n_sub x fast path + prologue/epilogue + the divergent blocks
that were taken; this is the number to watch while editing the kernel.
(Round 5, second sitting: the rare fall-back blocks now sit OUT OF LINE behind wave votes; the walker below follows the layout of the
fp32 instances, but splits the fp64 instance wrongly - for that one read registers / scratch here and take the instruction count from
the GPU: tools/valu_count.sh, profiles/<round>_counters.json.)"""
import os
import re
import subprocess
import sys

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(REPO, "marinevehiclereinforcementlearning_amd", "csrc")


def compile_s(extra, out, f64=False):
    sys.path.insert(0, REPO)
    from marinevehiclereinforcementlearning_amd import build
    build._gen_baked()
    flags = [f for f in (build.flags_for("gen/mvrl_rov6_f64.hip") if f64 else build.FLAGS) if f not in ("-fPIC",)]
    src = os.path.join(CSRC, "gen", "mvrl_rov6_f64.hip") if f64 else os.path.join(CSRC, "mvrl_rov6.hip")
    cmd = ["/opt/rocm/bin/hipcc"] + flags + list(extra) + ["-S", "--cuda-device-only", "-o", out, src]
    subprocess.run(cmd, check=True, stderr=subprocess.DEVNULL)


def analyse(path, sym, loop_depth=2):
    L = open(path).read().split("\n")
    start = next(i for i, l in enumerate(L) if l.startswith(sym + ":"))
    end = next(i for i in range(start, len(L)) if L[i].startswith(".Lfunc_end"))
    body = L[start:end]
    info = {}
    for l in L[end:end + 200]:
        m = re.match(r"\s*\.set .*\.(num_vgpr|num_agpr|numbered_sgpr|private_seg_size), (\d+)", l)
        if m:
            info.setdefault(m.group(1), int(m.group(2)))
    # walk the body: loop depth from the block labels' comments; lines behind `s_and_saveexec` + `s_cbranch_execz L` up to label L
    # are a divergent region (executed only by waves in which some lane took the branch)
    depth, skip_to, prev, seen_label = 0, None, "", False
    fast = outer = lds = 0
    slow, outer_div = [], 0
    for l in body:
        m = re.match(r"^(\.LBB\d+_\d+):\s*;?(.*)", l)
        if m:
            md = re.search(r"Depth[= ](\d)", m.group(2))
            depth = int(md.group(1)) if md else 0
            seen_label = True
            if skip_to == m.group(1):
                skip_to = None
            continue
        st = l.strip()
        mb = re.match(r"s_cbranch_execz (\.LBB\d+_\d+)", st)
        if mb and "saveexec" in prev and skip_to is None and seen_label:      # (the first one is the lane-range guard of the whole kernel)
            skip_to = mb.group(1)
            if depth >= loop_depth:
                slow.append(0)
        is_valu = bool(re.match(r"^\s+v_", l))
        if depth >= loop_depth:
            if skip_to is not None:
                if not slow:          # a divergent region entered outside the loop and still open inside it (out-of-line fall-back blocks)
                    slow.append(0)
                slow[-1] += is_valu
            else:
                fast += is_valu
                lds += bool(re.match(r"^\s+ds_", l))
        else:
            if skip_to is not None:
                outer_div += is_valu
            else:
                outer += is_valu
        if st and not st.startswith(";"):
            prev = st
    info["outer_divergent"] = outer_div
    return dict(fast=fast, slow=slow, outer=outer, lds=lds, **info)


if __name__ == "__main__":
    args = sys.argv[1:]
    zoh = "--zoh" in args
    flow = "--noflow" not in args
    fixed = "--fixed" in args
    f64 = "--f64" in args      # the fp64 twin (csrc/gen/mvrl_rov6_f64.hip, its own flags; single-step instance)
    extra = [a for a in args if a.startswith("-D") or a.startswith("-m") or a.startswith("-f")]
    out = "/tmp/isa/count64.s" if f64 else "/tmp/isa/count.s"
    os.makedirs("/tmp/isa", exist_ok=True)
    compile_s(extra, out, f64)
    b = lambda v: "Lb1E" if v else "Lb0E"
    ns = "_ZN6mvrl6416" if f64 else "_ZN4mvrl16"
    sym = ns + "rov6_step_kernelIPKNS_9Rov6BakedELb1E" + b(zoh) + b(flow) + "Li0E" + b(not f64) + b(fixed) + "EEvPKNS_7Rov6DevENS_6StepIOENS_7FlowDevE"
    r = analyse(out, sym, 1 if f64 else 2)   # the fp64 single-step instance has no fused-launch loop around the sub-step loop
    est = 4 * r["fast"] + r["outer"]
    print(f"sub-step loop: fast path {r['fast']} VALU, divergent blocks {r['slow']}, {r['lds']} LDS; outside the loop {r['outer']} (+{r['outer_divergent']} divergent: reset); "
          f"4 x fast + outside = {est}; vgpr {r.get('num_vgpr')} agpr {r.get('num_agpr')} sgpr {r.get('numbered_sgpr')} scratch {r.get('private_seg_size')}")
