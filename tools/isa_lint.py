"""ISA lint for the hazard of DESIGN 4h (round 4): a VMEM load with a multi-dword destination whose LAST destination register is read by a
VALU instruction within a few instructions of the `s_waitcnt vmcnt(0)` that covers it.  On gfx950 that read returned the register's OLD
contents in lanes 48-63 of one wave -- only in a workgroup that shared its CU with a lock-step twin of the same kernel
(rotary_attention_x3_kernel<72,4>, two workgroups per CU; tools/ubench/attn_hazard.hip).  The attention kernels now retire every prologue
load before the first use AND run one workgroup per CU; this lint looks for the same instruction pattern everywhere else:

    python tools/isa_lint.py            # compiles every csrc/*.hip to gfx950 assembly (-S, device only) and scans it
    python tools/isa_lint.py a.s b.s    # scans existing assembly

A hit = (kernel, line, load, the VALU instruction, distance).  Each kernel is listed with its register allocation: a kernel that needs more
than 256 registers per lane (one wave per SIMD) or whose launcher asks for more than half of the LDS cannot have a twin on its CU --
those hits are reported as `single` and do not count.  Exit status 1 when an uncounted hit is not in the allow-list below."""
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "rule-guided-music_amd", "csrc")
WINDOW = 3          # instructions behind the s_waitcnt in which a read of the last destination dword counts
# kernels whose launchers guarantee one workgroup per CU by LDS (common.h attn_prepare_kernel / attn_lds_one_per_cu)
ONE_PER_CU_BY_LDS = ("rotary_attention", "attn_bwd", "adaln_stream_kernel")
# (kernel substring, reason) pairs for hits that were read and argued.  The pattern is the commonest way to consume a
# vector load ("load x4 -> wait -> use") and on its own is no defect: the attention instance needed a lock-step twin on the same SIMDs AND
# went wrong in ~1 workgroup of 4000.  The families below run with such twins all the time (2-8 workgroups of the same kernel per CU);
# tools/hazard_soak.py runs them N times on fixed inputs and requires bit-identical results (profiles/r05_hazard_soak.txt: 0 differing
# runs of 100 per case, ~10^7 workgroup executions), and every parity test of the suite goes through them.  A NEW kernel with the
# pattern shows up as COUNTED until it is added to the soak and listed here.
SOAKED = "soaked (tools/hazard_soak.py)"
ALLOW = (("gemm_kernel", SOAKED), ("gemm2_kernel", SOAKED), ("gemm2_dual_kernel", SOAKED), ("gemm144_kernel", SOAKED),
         ("splitk_reduce", SOAKED), ("split_rows_kernel", SOAKED), ("ln_mod_kernel", SOAKED), ("ln_mod_bwd_kernel", SOAKED),
         ("gate_rows_kernel", SOAKED), ("ce_grad_kernel", "one 16-byte load per thread, consumed once; chord classifier path (golden tests)"),
         ("gn_", SOAKED), ("sumpool2_kernel", SOAKED), ("vae_conv_in", SOAKED), ("vae_conv_out", SOAKED),
         ("scg_rebuild_kernel", "the winner's rows copied through registers: bit-identity against the unsharded step is asserted in every SCG test"))

LOAD = re.compile(r"^\s*(global_load_dwordx[234]|buffer_load_dwordx[234]|flat_load_dwordx[234]|scratch_load_dwordx[234])\s+v\[(\d+):(\d+)\]")
WAIT0 = re.compile(r"^\s*s_waitcnt\s+.*vmcnt\(0\)")
WAITN = re.compile(r"^\s*s_waitcnt\s+.*vmcnt\((\d+)\)")
VALU = re.compile(r"^\s*(v_[a-z0-9_]+)\s+(.*)$")
REG = re.compile(r"v\[(\d+):(\d+)\]|v(\d+)")


def reads_reg(operands, reg):
    """does the operand string (destination first) read VGPR `reg`?"""
    parts = operands.split(",")
    for src in parts[1:]:
        for m in REG.finditer(src):
            if m.group(3) is not None:
                if int(m.group(3)) == reg:
                    return True
            elif int(m.group(1)) <= reg <= int(m.group(2)):
                return True
    return False


def scan(path):
    hits, kernels = [], {}
    kernel, pending, armed = None, [], []      # pending: loads not yet covered by a wait; armed: (last reg, load text, countdown)
    with open(path) as f:
        lines = f.read().split("\n")
    for ln, line in enumerate(lines, 1):
        m = re.match(r"^(_Z\w+):", line)
        if m:
            kernel, pending, armed = m.group(1), [], []
            continue
        m = re.match(r"^\s*\.amdhsa_kernel\s+(\S+)", line)
        if m:
            kernel = m.group(1)
        m = re.match(r"^\s*\.amdhsa_next_free_vgpr\s+(\d+)", line)
        if m and kernel:
            kernels.setdefault(kernel, {})["regs"] = int(m.group(1))
        if kernel is None or line.lstrip().startswith((";", ".")) or not line.strip():
            continue
        m = LOAD.match(line)
        if m:
            pending.append((int(m.group(3)), line.strip()))
            continue
        if WAIT0.match(line):
            armed = [(reg, txt, WINDOW) for reg, txt in pending]
            pending = []
            continue
        m = VALU.match(line)
        if armed:
            nxt = []
            for reg, txt, left in armed:
                if m and reads_reg(m.group(2), reg):
                    hits.append((kernel, ln, txt, line.strip(), WINDOW - left + 1))
                elif left > 1:
                    nxt.append((reg, txt, left - 1))
            armed = nxt
    return hits, kernels


def demangle(names):
    try:
        out = subprocess.run(["/opt/rocm/lib/llvm/bin/llvm-cxxfilt"], input="\n".join(names), capture_output=True, text=True).stdout.split("\n")
        return dict(zip(names, out))
    except Exception:
        return {n: n for n in names}


def main():
    files = sys.argv[1:]
    tmp = None
    if not files:
        tmp = tempfile.mkdtemp(prefix="isa_lint_")
        srcs = sorted(f for f in os.listdir(CSRC) if f.endswith(".hip"))
        procs = []
        for f in srcs:
            out = os.path.join(tmp, f[:-4] + ".s")
            procs.append((out, subprocess.Popen(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-S", "--cuda-device-only",
                                                 "-I" + CSRC, "-o", out, os.path.join(CSRC, f)], stderr=subprocess.DEVNULL)))
            if len(procs) % 8 == 0:
                for _, p in procs[-8:]:
                    p.wait()
        for out, p in procs:
            p.wait()
            files.append(out)
    total, counted = 0, 0
    for path in files:
        hits, kernels = scan(path)
        names = demangle(sorted({h[0] for h in hits}))
        by_kernel = {}
        for h in hits:
            by_kernel.setdefault(h[0], []).append(h)
        print(f"{os.path.basename(path)}: {len(kernels)} kernels, {len(hits)} pattern hits in {len(by_kernel)} kernels")
        for k, hs in sorted(by_kernel.items()):
            regs = kernels.get(k, {}).get("regs", 0)
            name = names.get(k, k)
            single = regs > 256 or any(s in name for s in ONE_PER_CU_BY_LDS)
            allowed = any(s in name for s, _ in ALLOW)
            total += len(hs)
            if not single and not allowed:
                counted += len(hs)
            tag = "single (one workgroup per CU: %s)" % ("registers" if regs > 256 else "LDS") if single else ("allowed" if allowed else "COUNTED")
            print(f"  {name[:150]}  [{regs} registers]  {len(hs)} hits  {tag}")
            for h in hs[:3]:
                print(f"      line {h[1]}: {h[2]}  ->  +{h[4]}: {h[3]}")
    print(f"total pattern hits {total}; counted (kernels that can share a CU with a twin, not argued) {counted}")
    return 1 if counted else 0


if __name__ == "__main__":
    sys.exit(main())
