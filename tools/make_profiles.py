"""Turn the output of tools/profile_round.sh (gpurun_out/pf) into the tracked summaries under profiles/.
usage: make_profiles.py <round-tag>   e.g. r2"""
import csv, glob, os, re, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PF = os.path.join(ROOT, "gpurun_out", "pf")
tag = sys.argv[1]


def short(name):
    name = name.replace("(anonymous namespace)::", "").replace("void ", "")
    m = re.match(r"^([A-Za-z0-9_:<>, ]+?)\(", name)
    name = m.group(1) if m else name
    if name.startswith("at::native"):
        name = "[torch] " + name
    return name[:80]


# 1. kernel statistics of the bench command
rows = list(csv.DictReader(open(os.path.join(PF, "kernel_stats.csv"))))
with open(os.path.join(ROOT, "profiles", f"{tag}_kernel_stats.csv"), "w") as f:
    f.write("# rocprofv3 --kernel-trace --stats -- python3 bench.py --steps 5 --warmup 2 --no-cpu --no-parity   "
            "(1 MI355X, 512 problems, N=2002; 7 evaluations of the timed path = kernels <false>, 6 of the optimiser path = kernels <true>)\n")
    f.write("Name,Calls,TotalDurationNs,AverageNs,Percentage,MinNs,MaxNs\n")
    for r in rows:
        f.write(f"{short(r['Name'])},{r['Calls']},{r['TotalDurationNs']},{float(r['AverageNs']):.0f},{r['Percentage']},{r['MinNs']},{r['MaxNs']}\n")


def counters(sub):
    out = {}
    # gpurun merges every run's files into gpurun_out/: only the newest collection of this pass counts
    files = sorted(glob.glob(os.path.join(PF, sub, "**", "*counter_collection.csv"), recursive=True), key=os.path.getmtime)
    for fn in files[-1:]:
        acc, disp = {}, {}
        for r in csv.DictReader(open(fn)):
            k = short(r["Kernel_Name"])
            acc.setdefault(k, {}).setdefault(r["Counter_Name"], 0.0)
            acc[k][r["Counter_Name"]] += float(r["Counter_Value"])
            disp.setdefault(k, set()).add(r["Dispatch_Id"])
        for k in acc:
            out[k] = {c: v / len(disp[k]) for c, v in acc[k].items()}
    return out


# 2. HBM traffic
fe, wr = counters("fetch"), counters("write")
units = 512 * 2002
alg = {"fwd_euler_sym_mfma40_kernel<false>": 26240, "esde_l96_ut_wave40_kernel<false, false, false>": 39688,
       "bwd_grad_euler_sym_mfma40_kernel<false, false, false>": 66240,
       # the optimiser path of bench.py: packed symmetric matrices (1088 kept entries in rows of 1152 doubles),
       # Psi(t) stored for t = 0 only
       "fwd_euler_sym_mfma40_kernel<true>": 8 * (1600 + 1088 + 80), "esde_l96_ut_wave40_kernel<false, true, true>": 8 * (1600 + 2 * 1088 + 161),
       "bwd_grad_euler_sym_mfma40_kernel<true, true, true>": 8 * (2 * 1600 + 2 * 1088 + 280)}
text = [f"# HBM traffic per launch from rocprofv3 PMC counters ({tag})",
        "# commands (two separate passes): rocprofv3 --pmc FETCH_SIZE --output-format csv -- python3 bench.py --steps 1 --warmup 1 --no-cpu --no-parity   (512 problems, the size bench.py times)",
        "#                                 rocprofv3 --pmc WRITE_SIZE --output-format csv -- python3 bench.py --steps 1 --warmup 1 --no-cpu --no-parity   (512 problems, the size bench.py times)",
        f"# workload: Lorenz-96 D=40 N=2002, 512 problems = {units} problem*timesteps per launch (averages over the launches of each kernel)",
        "# FETCH_SIZE / WRITE_SIZE are in KiB; on gfx950 FETCH_SIZE reports half of the bytes of 16-byte-per-lane",
        "# coalesced streaming reads (128-byte requests tallied as 64), so reads are doubled.",
        "# bench.py reads the LAST line below whose first field is the kernel it reports the roofline for.",
        "kernel,FETCH_SIZE_KiB,WRITE_SIZE_KiB,read_bytes(2x),write_bytes,total_bytes,bytes_per_unit,algorithmic_bytes_per_unit"]
for k, a in alg.items():
    if k not in fe or k not in wr:
        continue
    f_, w_ = fe[k]["FETCH_SIZE"], wr[k]["WRITE_SIZE"]
    rb, wb = 2 * 1024 * f_, 1024 * w_
    text.append(f"{k},{f_:.0f},{w_:.0f},{rb:.4g},{wb:.4g},{rb + wb:.4g},{(rb + wb) / units:.0f},{a}")
for name in (f"{tag}_pmc_hbm_traffic.txt", "pmc_hbm_traffic.txt"):
    open(os.path.join(ROOT, "profiles", name), "w").write("\n".join(text) + "\n")

# 3. issue / busy counters of the energy kernel
k = "esde_l96_ut_wave40_kernel<false, false, false>"
sq = dict(counters("sq_a")[k]); sq.update(counters("sq_b")[k])
steps = units
lines = [f"# rocprofv3 --pmc <counters> -- python3 tools/diag/esde_only.py 512   ({tag}; two passes; averages over 3 launches of {k},",
         f"# 512 problems x 2002 steps = {steps} wavefronts, one per step)",
         "# SQ_WAVE_CYCLES, SQ_WAIT_*, SQ_ACTIVE_INST_* count quad-cycles summed over wavefronts; SQ_VALU_MFMA_BUSY_CYCLES counts cycles",
         "counter,value,per_step"]
for c in sorted(sq):
    lines.append(f"{c},{sq[c]:.0f},{sq[c] / steps:.1f}")
wc = sq["SQ_WAVE_CYCLES"]
lines += ["# derived",
          f"# wavefront cycles per step: {4 * wc / steps:.0f}",
          f"# waiting on an instruction to issue or complete (SQ_WAIT_INST_ANY / SQ_WAVE_CYCLES): {sq['SQ_WAIT_INST_ANY'] / wc:.3f}",
          f"# waiting on memory counters (SQ_WAIT_ANY / SQ_WAVE_CYCLES): {sq['SQ_WAIT_ANY'] / wc:.3f}",
          f"# issuing (SQ_ACTIVE_INST_ANY / SQ_WAVE_CYCLES): {sq['SQ_ACTIVE_INST_ANY'] / wc:.3f}",
          f"# matrix-core instructions per step: {sq['SQ_INSTS_MFMA'] / steps:.1f}, busy cycles per instruction: {sq['SQ_VALU_MFMA_BUSY_CYCLES'] / sq['SQ_INSTS_MFMA']:.1f}",
          f"# vector instructions per step: {sq['SQ_INSTS_VALU'] / steps:.0f}; vector-ALU active cycles per step (4 x SQ_ACTIVE_INST_VALU): {4 * sq['SQ_ACTIVE_INST_VALU'] / steps:.0f}",
          f"# matrix-core busy cycles per step: {sq['SQ_VALU_MFMA_BUSY_CYCLES'] / steps:.0f}",
          "# two wavefronts share a SIMD: SIMD cycles per step = wavefront cycles per step / 2"]
open(os.path.join(ROOT, "profiles", f"{tag}_esde_sq_counters.txt"), "w").write("\n".join(lines) + "\n")
print("\n".join(lines[-9:]))
