"""The fp64 twin of the kernels is DERIVED TEXT (tools/gen_f64.py: float -> double, f-suffixed literals and libm names
widened).  A regex derivation can rot silently when the fp32 source grows a construct it does not know, so this checks,
on the CPU, that (1) the files under csrc/gen/ are exactly what the generator makes of the CURRENT fp32 sources, and
(2) nothing single-precision survives in the generated code: no `float` type, no f-suffixed literal, no *f libm call, no
fp32-only intrinsic - outside comments and outside blocks that are compiled out in the fp64 build."""
import os
import re
import sys

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(REPO, "tools"))
import gen_f64  # noqa: E402


def _strip_comments(text):
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return re.sub(r"//[^\n]*", "", text)


# macros that are only ever defined in the fp32 build (each is set under `#if !MVRL_F64` in the sources)
FP32_ONLY_MACROS = {"MVRL_PARK_ON", "MVRL_STAMP_ON", "MVRL_NATIVE_TRIG"}


def _drop_fp32_only_blocks(text):
    """Remove `#if !MVRL_F64 ... #endif/#else` regions and the `#else` side of `#if MVRL_F64` (crude nesting-aware pass)."""
    out, stack = [], []      # stack entries: True = currently emitting
    for line in text.splitlines():
        s = line.strip()
        if s.startswith("#if"):
            cond = s[3:].strip() if not s.startswith("#ifdef") and not s.startswith("#ifndef") else None
            if s.startswith("#ifdef") and s.split()[1] in FP32_ONLY_MACROS:
                stack.append(("f32", False))
            elif cond is not None and re.match(r"^!\s*MVRL_F64\b", cond):
                stack.append(("f32", False))
            elif cond is not None and re.match(r"^MVRL_F64\b", cond):
                stack.append(("f64", True))
            else:
                stack.append(("other", True))
            continue
        if s.startswith("#else") and stack:
            kind, emit = stack[-1]
            if kind == "f32":
                stack[-1] = (kind, True)
            elif kind == "f64":
                stack[-1] = (kind, False)
            continue
        if s.startswith("#elif") and stack:
            continue
        if s.startswith("#endif") and stack:
            stack.pop()
            continue
        if all(e for _, e in stack):
            out.append(line)
    return "\n".join(out)


def test_generated_sources_are_current():
    for f in gen_f64.FILES:
        src = os.path.join(gen_f64.CSRC, f)
        stem, ext = os.path.splitext(f)
        dst = os.path.join(gen_f64.GEN, f"{stem}_f64{ext}")
        if not os.path.exists(dst):
            gen_f64.main()
        assert open(dst).read() == gen_f64.transform(open(src).read(), f), f"{dst} is stale: run tools/gen_f64.py (build.py does)"


def test_nothing_single_precision_survives():
    for f in gen_f64.FILES:
        text = gen_f64.transform(open(os.path.join(gen_f64.CSRC, f)).read(), f)
        code = _drop_fp32_only_blocks(_strip_comments(text))
        assert not re.search(r"\bfloat\b", code), f
        assert not re.search(r"\bfloat[234]\b", code), (f, re.findall(r".*\bfloat[234]\b.*", code)[:3])
        lit = re.findall(r"(?<![\w.])(?:\d+\.\d*|\.\d+|\d+)(?:[eE][-+]?\d+)?f\b", code)
        assert not lit, (f, lit[:5])
        calls = re.findall(r"\b(?:fmaf|fabsf|floorf|rintf|sqrtf|expf|fminf|fmaxf|powf|sinf|cosf|sincosf|copysignf)\s*\(", code)
        assert not calls, (f, calls[:5])
        assert "__builtin_amdgcn_sinf" not in code and "__builtin_amdgcn_cosf" not in code, f
        assert "namespace mvrl64" in text or f.endswith(".hpp") and "mvrl64" in text, f
