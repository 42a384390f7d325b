"""Static instruction mix of the gfx950 kernels: compiles csrc/smc_filter.hip to assembly
(device only) and prints, per kernel whose mangled name contains one of the given substrings,
the number of instructions by class (fp64 VALU, 32-bit integer multiplies, other VALU, SALU,
LDS, global memory, barriers/waits) and the register / LDS budget from the kernel descriptor.

    python tools/isa_mix.py k_propagate k_ancestors2          # static counts, whole kernel body

The counts are static (every basic block once), so loops and rarely-taken slow paths count
once; for the straight-line step kernels that is the per-thread instruction count of the
common path plus the slow paths listed separately by label when --blocks is given.
"""
import collections
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "particles_amd", "csrc")


def compile_asm(src="smc_filter.hip", out="/tmp/isa_mix.s", extra=()):
    from particles_amd import _build
    flags = [f for f in _build.FLAGS if f not in ("-shared", "-fPIC")]
    cmd = [_build._hipcc()] + flags + ["--cuda-device-only", "-S", "-o", out,
                                       os.path.join(CSRC, src)] + list(extra)
    subprocess.run(cmd, check=True, stderr=subprocess.DEVNULL)
    return out


def classify(op):
    if op.startswith(("v_mul_lo_u32", "v_mul_hi_u32", "v_mad_u64_u32", "v_mad_i64_i32", "v_mul_hi_i32")):
        return "valu_imul"
    if op.startswith(("v_rcp_f64", "v_rsq_f64", "v_sqrt_f64", "v_div_scale_f64", "v_div_fmas_f64",
                      "v_div_fixup_f64")):
        return "valu_f64_div"
    if re.match(r"v_(fma|mul|add|max|min|fmac|ldexp|frexp|rndne|ceil|floor|trunc|fract|cmp\w*|cvt\w*)_\w*f64", op) \
            or op.endswith("_f64") or "_f64_" in op:
        return "valu_f64"
    if op.startswith("v_mfma"):
        return "mfma"
    if op.startswith("v_"):
        return "valu_other"
    if op.startswith("s_waitcnt") or op.startswith("s_barrier") or op.startswith("s_nop") or op.startswith("s_sleep"):
        return "wait/barrier"
    if op.startswith("s_load") or op.startswith("s_buffer"):
        return "smem"
    if op.startswith("s_"):
        return "salu"
    if op.startswith("ds_"):
        return "lds"
    if op.startswith(("global_", "flat_", "buffer_", "scratch_")):
        return "vmem"
    return "other"


def kernels(path):
    cur, body, out = None, [], {}
    meta = {}
    for line in open(path):
        m = re.match(r"^(_Z\w+):", line)
        if m and ".kd" not in line:
            cur, body = m.group(1), []
            out[cur] = body
            continue
        if cur and re.match(r"^\s+\.(end_amdhsa_kernel|section|size)", line) is not None and ".size" in line:
            cur = None
            continue
        m = re.match(r"^\s+\.amdhsa_(next_free_vgpr|next_free_sgpr|group_segment_fixed_size|accum_offset)\s+(\S+)", line)
        if m:
            meta.setdefault("_pending", {})[m.group(1)] = m.group(2)
        m = re.match(r"^\s+\.amdhsa_kernel\s+(\S+)", line)
        if m:
            meta["_name"] = m.group(1)
            meta["_pending"] = {}
        if re.match(r"^\s+\.end_amdhsa_kernel", line) and "_name" in meta:
            meta[meta.pop("_name")] = meta.pop("_pending")
        if cur is not None:
            s = line.strip()
            if not s or s.startswith((";", ".", "//")) or s.endswith(":"):
                if s.endswith(":") and not s.startswith("."):
                    pass
                if re.match(r"^\.LBB\d+_\d+:", s):
                    body.append(("label", s))
                continue
            body.append(("ins", s.split()[0]))
    return out, meta


def main():
    pats = [a for a in sys.argv[1:] if not a.startswith("--")]
    path = "/tmp/isa_mix.s"
    if "--reuse" not in sys.argv or not os.path.exists(path):
        sys.path.insert(0, ROOT)
        compile_asm(out=path)
    ks, meta = kernels(path)
    for name, body in ks.items():
        if pats and not any(p in name for p in pats):
            continue
        c = collections.Counter(classify(op) for kind, op in body if kind == "ins")
        tot = sum(c.values())
        if tot < 20:
            continue
        md = meta.get(name, {})
        print("%s\n   total %d | %s | vgpr %s sgpr %s lds %s" % (
            name, tot, " ".join("%s %d" % kv for kv in sorted(c.items(), key=lambda kv: -kv[1])),
            md.get("next_free_vgpr"), md.get("next_free_sgpr"), md.get("group_segment_fixed_size")))
        if "--ops" in sys.argv:
            oc = collections.Counter(op for kind, op in body if kind == "ins")
            print("   " + ", ".join("%s %d" % kv for kv in oc.most_common(40)))


if __name__ == "__main__":
    main()
