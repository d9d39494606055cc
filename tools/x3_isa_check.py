#!/usr/bin/env python3
"""Build-time check of the split-operand kernels' ISA (run by build_lib.sh on the compiler's own .s of convx.hip / wgradx.hip).

Round 5 met a timing-dependent staging fault in conv_x3_kernel<3,*,true,true,*,*> that appeared with the SLP vectoriser's packed
fp32 arithmetic (v_pk_mul_f32 ... op_sel) in the patch staging and went away with scalar multiplies; its cause was never
established (round 6: the wait accounting of both code shapes is identical instruction for instruction, and the packed shape no
longer reproduces the fault -- DESIGN.md section 4).  Until it is explained the hazard is fenced off structurally: both sources
are compiled with -fno-slp-vectorize, and this check fails the build if any packed fp32 VALU instruction appears in a
conv_x3_kernel / conv_wgradx_kernel body all the same.  It also verifies the round-6 barrier form: no compiler-generated
`s_waitcnt vmcnt(0)` directly in front of an s_barrier of conv_x3's stage loop other than the hand-written ones (a
__syncthreads() there makes every chunk wait out its ring DMA at issue).

usage: tools/x3_isa_check.py <convx.s> [<wgradx.s>]      exit status 0 = clean"""
import re
import sys

PACKED = re.compile(r"\bv_pk_(mul|fma|add)_f32\b")


def functions(path):
    out, name = {}, None
    for line in open(path, errors="replace"):
        m = re.match(r"^(_Z\w+):", line)
        if m:
            name = m.group(1); out[name] = []
        elif name is not None:
            out[name].append(line)
            if "s_endpgm" in line:
                name = None
    return out


def main(paths):
    bad = 0
    seen = 0
    for path in paths:
        for name, body in functions(path).items():
            if "conv_x3_kernel" not in name and "conv_wgradx_kernel" not in name:
                continue
            seen += 1
            packed = [l.strip() for l in body if PACKED.search(l)]
            if packed:
                bad += 1
                print("x3_isa_check: %s: %d packed fp32 instructions (first: %s)" % (name, len(packed), packed[0]))
            if "conv_x3_kernel" in name:
                # barriers with a compiler-generated full wait in front: only the wide epilogue's __syncthreads() may have one (every
                # DMA has been waited for by hand by then); the stage loop's barriers are bare (CX_BARRIER)
                raw = [l.strip() for l in body if l.strip()]
                n_auto = sum(1 for i, l in enumerate(raw) if l.startswith("s_barrier") and i > 1 and raw[i - 1].startswith("s_waitcnt vmcnt(0) lgkmcnt(0)"))
                if n_auto > 1:
                    bad += 1
                    print("x3_isa_check: %s: %d barriers carry a compiler-generated 's_waitcnt vmcnt(0) lgkmcnt(0)' (stage loop uses __syncthreads()?)" % (name, n_auto))
    if seen == 0:
        print("x3_isa_check: no conv_x3_kernel / conv_wgradx_kernel bodies found in %s" % " ".join(paths))
        return 2
    print("x3_isa_check: %d kernel bodies, %s" % (seen, "clean" if not bad else "%d with findings" % bad))
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main(sys.argv[1:]))
