#!/usr/bin/env python3
"""Static issue-dependency scan of a kernel's ISA (hipcc -S output): for every VALU instruction, the distance (in VALU
instructions of the same wave) to the producer of its most recent source operand.  On gfx950 a wave64 VALU instruction
occupies a SIMD-32 for 2 cycles, but a consumer that directly follows its producer cannot issue for ~4 (tools/valu_peak.hip:
dependent chains sustain 0.49 wave-instr/ns/SIMD at any occupancy against 1.03 for independent ones), so the share of
distance-1 (and, mildly, distance-2..4) pairs is the schedule's quality figure.

    python tools/ilp_scan.py file.s kernel_substring [first_line last_line]
"""
import re
import sys

REG = re.compile(r"\b([vs])(\d+)\b|\b([vs])\[(\d+):(\d+)\]|\b(vcc|exec|scc)\b")


def regs(tok):
    out = set()
    for m in REG.finditer(tok):
        if m.group(1):
            out.add((m.group(1), int(m.group(2))))
        elif m.group(3):
            for k in range(int(m.group(4)), int(m.group(5)) + 1):
                out.add((m.group(3), k))
        elif m.group(6):
            out.add((m.group(6), 0))
    return out


def scan(lines):
    last_writer = {}
    hist = {}
    n = 0
    idx = 0
    for ln in lines:
        ln = ln.split(";")[0].strip()
        if not ln or ln.endswith(":") or ln.startswith("."):
            continue
        parts = ln.split(None, 1)
        op = parts[0]
        ops = [t.strip() for t in parts[1].split(",")] if len(parts) > 1 else []
        if not op.startswith("v_"):
            # scalar / memory instructions: track their writes so VALU readers see them, but they issue on other ports
            if ops and (op.startswith("s_") or op.startswith("global_load") or op.startswith("ds_read")):
                for r in regs(ops[0]):
                    last_writer[r] = -10 ** 9
            continue
        idx += 1
        n += 1
        if op.startswith("v_cmp"):
            if op.endswith("_e64") and ops:
                dst, srcs = regs(ops[0]), ops[1:]
            else:
                dst, srcs = {("vcc", 0)}, ops
        else:
            dst, srcs = (regs(ops[0]) if ops else set()), ops[1:]
            if op.startswith("v_fmac") or op.startswith("v_mac"):
                srcs = ops  # accumulates into dst
            if op.startswith("v_cndmask") and op.endswith("_e32"):
                srcs = srcs + ["vcc"]
        d = None
        for t in srcs:
            for r in regs(t):
                w = last_writer.get(r)
                if w is not None and w > 0:
                    dd = idx - w
                    d = dd if d is None else min(d, dd)
        key = d if d is not None and d <= 8 else ">8"
        hist[key] = hist.get(key, 0) + 1
        for r in dst:
            last_writer[r] = idx
    return n, hist


def main():
    path, name = sys.argv[1], sys.argv[2]
    src = open(path).read().splitlines()
    start = next(i for i, l in enumerate(src) if l.startswith("_Z") and name in l and l.rstrip().endswith(":") or (name in l and l.startswith("_Z") and ":" in l))
    end = next(i for i in range(start + 1, len(src)) if src[i].strip().startswith("s_endpgm"))
    body = src[start:end]
    if len(sys.argv) >= 5:
        body = body[int(sys.argv[3]):int(sys.argv[4])]
    n, hist = scan(body)
    print(f"{n} VALU instructions")
    cost = 0.0
    for k in sorted(hist, key=lambda x: (isinstance(x, str), x)):
        print(f"  producer distance {k}: {hist[k]} ({100.0 * hist[k] / n:.1f} %)")
    # issue model from the calibration: distance 1 -> 4 cycles, distance 2..4 -> ~2.2, else 2
    cyc = sum(v * (4.0 if k == 1 else (2.2 if k in (2, 3, 4) else 2.0)) for k, v in hist.items())
    print(f"  modelled issue cycles/instr {cyc / n:.2f}  (2.00 = peak)")


if __name__ == "__main__":
    main()
