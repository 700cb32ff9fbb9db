#!/usr/bin/env python3
"""Developer tool: the ISA of a run-time compiled kernel form of a script-defined model, without a GPU.

    python3 tools/rtc_isa.py <workload> [instantiation]      workload: vec | vec_auto | vec1 | vec1_auto | mala
prints the instruction mix of the kernel's sampling loop region (whole kernel) and writes the listing to /tmp/rtc_isa_<workload>.s.
amc_model_check builds the form (AMC_MODEL_CHECK_INST), AMC_RTC_CACHE_DIR keeps the code object, llvm-objdump reads it."""
import collections, glob, os, re, struct, subprocess, sys, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
DRIFT = ("theta0 + theta1*z", "-((delta-theta0)*(delta-theta0))/(2.0*theta1*theta1) - amc_log(theta1)",
         ["(delta-theta0)/(theta1*theta1)", "((delta-theta0)*(delta-theta0))/(theta1*theta1*theta1) - 1.0/theta1"])
GAUSS = ("sigma*z", "-(delta*delta)/(2.0*(sigma*sigma)) - amc_log(6.283185307179586*(sigma*sigma))/2.0", "(delta*delta)/(sigma*sigma*sigma) - 1.0/sigma")
MALA = ("-2.0*sigma*sigma*x + sigma*z", "-((delta + 2.0*sigma*sigma*x)*(delta + 2.0*sigma*sigma*x))/(2.0*(sigma*sigma)) - amc_log(sigma)",
        "((delta + 2.0*sigma*sigma*x)*(delta + 2.0*sigma*sigma*x))/(sigma*sigma*sigma) - 4.0*x*(delta + 2.0*sigma*sigma*x)/sigma - 1.0/sigma")
wl = sys.argv[1]
inst = sys.argv[2] if len(sys.argv) > 2 else "amc::pg_estimate_kernel<2,1,false,1,0,false>"
cache = tempfile.mkdtemp(prefix="rtc_isa_")
os.environ["AMC_RTC_CACHE_DIR"] = cache
os.environ["AMC_MODEL_CHECK_INST"] = inst
from montecarlo_amd import _capi as A
kw = {"vec": dict(sample=DRIFT[0], logq=DRIFT[1], dlogq=DRIFT[2], n_params=2), "vec_auto": dict(sample=DRIFT[0], logq=DRIFT[1], n_params=2),
      "vec1": dict(sample=GAUSS[0], logq=GAUSS[1], dlogq=GAUSS[2]), "vec1_auto": dict(sample=GAUSS[0], logq=GAUSS[1]),
      "mala": dict(sample=MALA[0], logq=MALA[1], dlogq=MALA[2]),
      "mixed": dict(sample=[GAUSS[0], MALA[0]], logq=[GAUSS[1], MALA[1]], dlogq=[GAUSS[2], MALA[2]]),
      "mixed_auto": dict(sample=[GAUSS[0], MALA[0]], logq=[GAUSS[1], MALA[1]], dlogq=[GAUSS[2], None])}[wl]
A.model_check(**kw)
f = glob.glob(os.path.join(cache, "*.bin"))[0]
raw = open(f, "rb").read()
magic, nl, cl = struct.unpack("<QQQ", raw[:24])
co = os.path.join(cache, "k.co")
open(co, "wb").write(raw[24 + nl:24 + nl + cl])
out = f"/tmp/rtc_isa_{wl}.s"
dis = subprocess.run(["/opt/rocm/lib/llvm/bin/llvm-objdump", "-d", "--mcpu=gfx950", co], capture_output=True, text=True).stdout
open(out, "w").write(dis)
# the kernel's own function: from its label to the next one
lines = dis.splitlines()
want = inst.split("<")[0].split("::")[-1]
start = next(i for i, l in enumerate(lines) if re.match(r"^[0-9a-f]+ <.*" + want + r"I?", l))
end = next((i for i in range(start + 1, len(lines)) if re.match(r"^[0-9a-f]+ <", lines[i]) and not re.match(r"^[0-9a-f]+ <L\d", lines[i])), len(lines))
body = []          # (address, mnemonic, text)
for l in lines[start + 1:end]:
    m = re.match(r"\s+([a-z_0-9]+)\s*(.*?)\s*//\s*([0-9A-F]+):", l)
    if m:
        body.append((int(m.group(3), 16), m.group(1), m.group(2)))
addr_index = {a: i for i, (a, _, _) in enumerate(body)}
# loops: backward branches; the trip of interest is the LARGEST loop body that contains no other backward branch target beyond itself
loops = []
for i, (a, op, text) in enumerate(body):
    if op.startswith("s_cbranch") or op == "s_branch":
        m = re.search(r"(-?\d+)", text)
        if not m:
            continue
        off = int(m.group(1))
        if off >= 32768:
            off -= 65536
        tgt = a + 4 + 4 * off
        if off < 0 and tgt in addr_index:
            loops.append((addr_index[tgt], i))
def mix(lo, hi):
    c = collections.Counter(op for _, op, _ in body[lo:hi + 1])
    return c
print(f"{wl} {inst}: {len(body)} instructions in the kernel, {sum(1 for _, o, _ in body if o.startswith('v_'))} vector ALU; listing {out}")
meta = subprocess.run(["/opt/rocm/lib/llvm/bin/llvm-readelf", "--notes", co], capture_output=True, text=True).stdout
blk = meta[meta.index(want):] if want in meta else meta
for key in (".vgpr_count", ".sgpr_count", ".vgpr_spill_count", ".sgpr_spill_count", ".agpr_count", ".group_segment_fixed_size"):
    m = re.search(re.escape(key) + r":\s+(\d+)", blk)
    if m:
        print(f"  {key[1:]} = {m.group(1)}")
# the trips of the sampling loop: loops that hold the Philox rounds (v_mad_u64_u32) and a global load; innermost first
seen = set()
for lo, hi in sorted(loops, key=lambda t: t[1] - t[0]):
    c = mix(lo, hi)
    if c.get("v_mad_u64_u32", 0) < 20 or (lo, hi) in seen:
        continue
    seen.add((lo, hi))
    nv = sum(n for o, n in c.items() if o.startswith("v_"))
    print(f"  loop of {hi - lo + 1} instructions at {body[lo][0]:#x}: {nv} vector ALU, {sum(n for o, n in c.items() if o.startswith('s_'))} scalar; "
          f"{c.get('v_div_scale_f64', 0)} v_div_scale_f64, {c.get('v_rcp_f64_e32', 0)} v_rcp_f64, {c.get('v_readlane_b32', 0)} v_readlane, "
          f"{c.get('v_writelane_b32', 0)} v_writelane, {c.get('v_mad_u64_u32', 0)} v_mad_u64_u32, {sum(n for o, n in c.items() if o.startswith('ds_'))} LDS, "
          f"{sum(n for o, n in c.items() if o.startswith('global_') or o.startswith('buffer_'))} memory")
    if len(seen) == 1 or os.environ.get("ALL_LOOPS"):
        print("     " + ", ".join(f"{n} {o}" for o, n in c.most_common(40)))
