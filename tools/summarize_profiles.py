"""Turn the raw rocprofv3 output of tools/profile_round.sh (gpurun_out/<tag>_*) into the committed
summaries under profiles/<out>_*.   usage: python tools/summarize_profiles.py r01b r01"""
import collections, csv, glob, json, os, shutil, sys

R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag, out = sys.argv[1], sys.argv[2]
G, P = os.path.join(R, "gpurun_out"), os.path.join(R, "profiles")


def one(pattern):
    f = sorted(glob.glob(os.path.join(G, pattern)), key=os.path.getmtime)
    assert f, pattern
    return f[-1]  # the newest run of that pass


def short(name):
    """kernel name without its trailing argument list"""
    name = name.strip()
    if not name.endswith(")"):
        return name
    depth = 0
    for i in range(len(name) - 1, -1, -1):
        depth += name[i] == ")"
        depth -= name[i] == "("
        if depth == 0:
            return name[:i]
    return name


for src, dst in [("%s_infer608_b64_bench.json", "%s_infer608_b64_bench.json"),
                 ("%s_infer608_b64_bench_under_rocprof.json", "%s_infer608_b64_bench_under_rocprof.json"),
                 ("%s_train416_b16_bench.json", "%s_train416_b16_bench.json"),
                 ("%s_train416_b16_bench_under_rocprof.json", "%s_train416_b16_bench_under_rocprof.json"),
                 ("%s_layers_608_b64.txt", "%s_layers_608_b64.txt"),
                 ("%s_layers_608_b64_split.txt", "%s_layers_608_b64_split.txt"),
                 ("%s_infer608_b64_split_bench_under_rocprof.json", "%s_infer608_b64_split_bench_under_rocprof.json"),
                 ("%s_small_batch_latency.txt", "%s_small_batch_latency.txt"),
                 ("%s_layers_608_b1.txt", "%s_layers_608_b1.txt"),
                 ("%s_layers_608_b1_split.txt", "%s_layers_608_b1_split.txt"),
                 ("%s_ab_split_train.txt", "%s_ab_split_train.txt"),
                 ("%s_train416_b16_split_kernel_totals.txt", "%s_train416_b16_split_kernel_totals.txt"),
                 ("%s_nms_latency.txt", "%s_nms_latency.txt"),
                 ("%s_small_batch_latency_split.txt", "%s_small_batch_latency_split.txt"),
                 ("%s_layers.txt", "%s_train416_b16_layers.txt")]:
    if os.path.exists(os.path.join(G, src % tag)):
        shutil.copy(os.path.join(G, src % tag), os.path.join(P, dst % out))
shutil.copy(one("%s_p_inf/*/*kernel_stats.csv" % tag), os.path.join(P, "%s_infer608_b64_kernel_stats.csv" % out))
shutil.copy(one("%s_p_trn/*/*kernel_stats.csv" % tag), os.path.join(P, "%s_train416_b16_kernel_stats.csv" % out))
if glob.glob(os.path.join(G, "%s_p_spl/*/*kernel_stats.csv" % tag)):
    shutil.copy(one("%s_p_spl/*/*kernel_stats.csv" % tag), os.path.join(P, "%s_infer608_b64_split_kernel_stats.csv" % out))


def counters(pattern):
    agg = collections.defaultdict(lambda: collections.defaultdict(float))
    disp = collections.defaultdict(set)
    for r in csv.DictReader(open(one(pattern))):
        k = short(r["Kernel_Name"])
        agg[k][r["Counter_Name"]] += float(r["Counter_Value"])
        disp[k].add(r["Dispatch_Id"])
    return agg, {k: len(v) for k, v in disp.items()}


WORK = {"inf": ("infer608_b64", "`python3 bench.py --steps 1 --warmup 1 --cpu-frames 0 --no-roofline --no-pmc --no-latency` (608x608, batch 64)"),
        "trn": ("train416_b16", "`python3 bench.py --mode train --steps 1 --warmup 1 --no-roofline --no-pmc` (416x416, batch 16; 2 steps)"),
        "spl": ("infer608_b64_split", "`python3 bench.py --conv-mode split_bf16x3 --steps 1 --warmup 1 --cpu-frames 0 --no-roofline --no-pmc --no-latency` "
                                      "(608x608, batch 64, opt-in split-fp32 conv mode)")}
for m, (wname, cmd) in WORK.items():
    if not glob.glob(os.path.join(G, "%s_%s_fetch/*/*counter_collection.csv" % (tag, m))):
        continue
    # HBM traffic: FETCH_SIZE / WRITE_SIZE from their own passes (KiB per dispatch); gfx950 tallies 128-B read
    # requests at 64 B, so FETCH_SIZE is doubled (MI355X_MICROARCH.md, HBM / rocprofv3 section)
    fetch, nf = counters("%s_%s_fetch/*/*counter_collection.csv" % (tag, m))
    write, nw = counters("%s_%s_write/*/*counter_collection.csv" % (tag, m))
    hbm = {"note": "rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE (two separate passes, no tracing combined) of %s. "
                   "Counters are KiB per dispatch; fetch_corrected doubles FETCH_SIZE per MI355X_MICROARCH.md section HBM "
                   "(gfx950 tallies 128-B read requests at 64 B); WRITE_SIZE is taken as is." % cmd, "kernels": {}}
    for k in fetch:
        if k not in write or "at::native" in k:
            continue
        f = fetch[k]["FETCH_SIZE"] * 1024 / nf[k] / 1e6
        w = write[k]["WRITE_SIZE"] * 1024 / nw[k] / 1e6
        hbm["kernels"][k] = {"dispatches": nf[k], "fetch_MB_per_launch_raw": f, "fetch_MB_per_launch_corrected": 2 * f,
                             "write_MB_per_launch": w, "hbm_MB_per_launch": 2 * f + w,
                             "hbm_MB_total": (2 * f + w) * nf[k]}
    json.dump(hbm, open(os.path.join(P, "%s_%s_pmc_hbm.json" % (out, wname)), "w"), indent=1)
    # MFMA pipe occupancy: SQ_VALU_MFMA_BUSY_CYCLES counts busy cycles per SIMD; GRBM_GUI_ACTIVE is summed over
    # the 8 XCDs, each with 32 CUs x 4 SIMDs = 128 SIMDs
    mf, nm = counters("%s_%s_mfma/*/*counter_collection.csv" % (tag, m))
    mfma = {"note": "rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_WAVE_CYCLES SQ_INSTS_MFMA SQ_INSTS_VALU "
                    "SQ_INSTS_SALU SQ_INSTS_LDS GRBM_GUI_ACTIVE of the same command (own pass). mfma_busy_frac = "
                    "SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE x 128): GRBM_GUI_ACTIVE is summed over the 8 XCDs, each "
                    "has 128 SIMDs. cu_busy = SQ_BUSY_CU_CYCLES / GRBM_GUI_ACTIVE / 32 CUs per XCD.", "kernels": {}}
    for k, v in mf.items():
        if "at::native" in k or not v.get("GRBM_GUI_ACTIVE"):
            continue
        g = v["GRBM_GUI_ACTIVE"]
        mfma["kernels"][k] = {"dispatches": nm[k], "mfma_busy_frac": v["SQ_VALU_MFMA_BUSY_CYCLES"] / (g * 128),
                              "cu_busy_frac": v["SQ_BUSY_CU_CYCLES"] / g / 32,
                              "insts_per_launch": {c[9:].lower(): v[c] / nm[k] for c in v if c.startswith("SQ_INSTS_")}}
    json.dump(mfma, open(os.path.join(P, "%s_%s_pmc_mfma.json" % (out, wname)), "w"), indent=1)
    print("====", wname)
    for k, v in mfma["kernels"].items():
        print("%-60s mfma busy %.3f  cu busy %.3f" % (k[:60], v["mfma_busy_frac"], v["cu_busy_frac"]))
    for k, v in hbm["kernels"].items():
        print("%-60s HBM %.1f MB/launch x %d" % (k[:60], v["hbm_MB_per_launch"], v["dispatches"]))
