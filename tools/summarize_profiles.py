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
                 ("%s_small_batch_latency_416.txt", "%s_small_batch_latency_416.txt"),
                 ("%s_layers_416_b1.txt", "%s_layers_416_b1.txt"),
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


# ---- profiles/<out>_summary.md: every number table of the round, generated from the committed JSON / CSV files above
# (DESIGN.md quotes this file; nothing in it is typed by hand) -----------------------------------------------------------
def _load(name):
    f = os.path.join(P, name)
    if not os.path.exists(f):
        return None
    raw = open(f).read().strip()
    try:
        return json.loads(raw)
    except ValueError:
        pass
    txt = raw.splitlines()   # (a bench output with library chatter before its one line)
    for line in reversed(txt):
        if line.startswith("{"):
            return json.loads(line)
    return None


def _f(v, fmt="%.1f"):
    return "—" if v is None else fmt % v


def write_summary():
    b = _load("%s_infer608_b64_bench.json" % out)
    t = _load("%s_train416_b16_bench.json" % out)
    if not b:
        return
    L = ["# Round summary `%s` (generated by tools/summarize_profiles.py from the files in this directory)" % out, "",
         "All numbers: ONE MI355X, one box (boxes of the pool differ by a few percent; ratios are same-run).", "",
         "## Inference (`%s_infer608_b64_bench.json`: `python bench.py`)" % out, "",
         "| leg | frames/s | ms/step | note |", "|---|---|---|---|"]
    r = b.get("roofline", {})
    L.append("| **headline**: 608², batch 64, 20 classes, exact fp32, frames resident | **%s** | %s | dominant kernel `%s`: %s TF = **%s** of the fp32 MFMA roof (%s launches, %s ms each); all conv launches %s; whole step %s TF; traffic %s GB per dominant launch (%s× algorithmic) |"
             % (_f(b["value"]), _f(b["ms_per_step"], "%.2f"), r.get("kernel"), _f(r.get("achieved")), _f(r.get("frac"), "%.3f"),
                r.get("launches_per_step"), _f(r.get("avg_launch_ms"), "%.3f"), _f(r.get("frac_all_conv"), "%.3f"),
                _f(r.get("whole_step_tflops")), _f((r.get("traffic") or 0) / 1e9, "%.2f") if r.get("traffic") else "—",
                _f((r.get("traffic_detail") or {}).get("over_algorithmic"), "%.2f")))
    a4 = b.get("also_416")
    if a4:
        L.append("| 416², batch 64 (`also_416`) | %s | %s | %s of the roof |" % (_f(a4["frames_per_s"]), _f(a4["ms_per_step"], "%.2f"), _f(a4.get("frac_of_fp32_mfma_peak"), "%.3f")))
    lb = b.get("latency_batch1")
    if lb:
        L.append("| one frame 608² (`latency_batch1`) | %s | %s | eager; HIP graph %s ms; %s of the roof |"
                 % (_f(1e3 / lb["eager_ms"]), _f(lb["eager_ms"], "%.3f"), _f(lb["hip_graph_ms"], "%.3f"), _f(lb.get("frac_of_fp32_mfma_peak"), "%.2f")))
    lb4 = b.get("latency_batch1_416")
    if lb4:
        L.append("| one frame 416² (`latency_batch1_416`: the reference's default detect call) | %s | %s | eager; HIP graph %s ms; %s of the roof |"
                 % (_f(1e3 / lb4["eager_ms"]), _f(lb4["eager_ms"], "%.3f"), _f(lb4["hip_graph_ms"], "%.3f"), _f(lb4.get("frac_of_fp32_mfma_peak"), "%.2f")))
    for key, label in (("also_hostfed608", "host-fed, pipelined (`also_hostfed608`)"), ("also_vid608", "configs[3]: 30 classes, host clip batch → scatter → net → gather (`also_vid608`)")):
        h = b.get(key)
        if h:
            L.append("| %s | %s | %s | source %s; copy-in alone %s ms (%s GB/s), copy-out %s ms%s |"
                     % (label, _f(h["frames_per_s"]), _f(h["ms_per_step"], "%.2f"), h["source_frames"], _f(h["copy_in_alone_ms"], "%.2f"),
                        _f(h["copy_in_GBps"]), _f(h["copy_out_alone_ms"], "%.3f"),
                        "; %s of the resident rate, %s ms per step exposed" % (_f(h["vs_resident"], "%.3f"), _f(h["exposed_ms_per_step"], "%.2f")) if "vs_resident" in h else ""))
    s = b.get("also_infer608_split")
    if s:
        sr = s.get("roofline", {})
        L.append("| opt-in conv mode `split_bf16x3` (`also_infer608_split`; NOT the parity path) | %s | %s | %s× the exact step; split + Winograd launches %s TF-equivalent = %s of the bf16 roof ÷ 6 (%s of the fp32 roof), bf16 issued %s TF; Winograd %s launches %s ms; traffic %s GB per launch (%s×); one frame %s ms |"
                 % (_f(s["frames_per_s"]), _f(s["ms_per_step"], "%.2f"), _f(s["speedup_over_exact"], "%.3f"), _f(sr.get("achieved_fp32_equivalent")),
                    _f(sr.get("frac_vs_bf16_peak_over_6_419"), "%.3f"), _f(sr.get("frac_vs_fp32_mfma_peak_157"), "%.2f"), _f(sr.get("bf16_mfma_tflops"), "%.0f"),
                    sr.get("winograd_launches"), _f(sr.get("winograd_ms"), "%.2f"), _f((sr.get("traffic") or 0) / 1e9, "%.2f") if sr.get("traffic") else "—",
                    _f((sr.get("traffic_detail") or {}).get("over_algorithmic"), "%.2f"), _f((s.get("latency_batch1") or {}).get("eager_ms"), "%.3f")))
    for key in ("cpu_baseline", "cpu_baseline_torch"):
        c = b.get(key)
        if c and c.get("value"):
            L.append("| `%s` (%s, %d host threads; NOT MXNet) | %s | — | %s |" % (key, c["kind"], c["cores"], _f(c["value"], "%.2f"), c["sample"][:110]))
    L += ["", "## Training (`also_train416*` of the same line; `%s_train416_b16_bench.json`: `python bench.py --mode train`)" % out, "",
          "| leg | frames/s | ms/step | of the fp32 roof | forward / backward ms | traffic GB/step |", "|---|---|---|---|---|---|"]
    for key, label in (("also_train416", "configs[2]: 416², 16 per GPU, exact"), ("also_train416_split", "the same, conv mode `split_bf16x3_train`"), ("also_syncbn608", "configs[4] (N > 1 only)")):
        g = b.get(key)
        if g:
            L.append("| %s (`%s`) | %s | %s | %s | %s / %s | %s |" % (label, key, _f(g["frames_per_s"]), _f(g["ms_per_step"], "%.2f"), _f(g.get("frac"), "%.3f"),
                                                                  _f(g.get("forward_ms"), "%.2f"), _f(g.get("backward_ms"), "%.2f"),
                                                                  _f(g["traffic"] / 1e9) if g.get("traffic") else "—"))
    ms = b.get("also_train_multiscale")
    if ms:
        L.append("| the reference's DEFAULT mode: sizes %s, %d steps each (`also_train_multiscale`) | %s | %s | %s | — | — |"
                 % ("/".join(str(v) for v in ms["sizes_in_order"]), ms["interval"], _f(ms["frames_per_s"]), _f(ms["ms_per_step"], "%.2f"), _f(ms.get("frac"), "%.3f")))
    if t:
        L.append("| `--mode train` as the headline | %s | %s | %s | %s / %s | %s |"
                 % (_f(t["value"]), _f(t["ms_per_step"], "%.2f"), _f((t.get("roofline") or {}).get("frac"), "%.3f"),
                    _f((t.get("step_split") or {}).get("forward_ms"), "%.2f"), _f((t.get("step_split") or {}).get("backward_ms"), "%.2f"),
                    _f((t.get("roofline") or {}).get("traffic", 0) / 1e9) if (t.get("roofline") or {}).get("traffic") else "—"))
    if ms:
        L += ["", "### Multi-scale training, per size (rank 0; re-plan = workspace bind + border zeroing at every size change; %.2f %% of the sequence)"
              % (100 * ms["replan_share"]), "", "| size | frames/s | steady ms/step | whole-step fraction of the roof | re-plan ms | share of its 10-step window |", "|---|---|---|---|---|---|"]
        for sz in sorted(ms["per_size"], key=int):
            v = ms["per_size"][sz]
            L.append("| %s | %s | %s | %s | %s | %s %% |" % (sz, _f(v["frames_per_s"]), _f(v["steady_ms_per_step"], "%.2f"), _f(v["frac_whole_step"], "%.3f"),
                                                          _f(v["replan_ms"], "%.2f"), _f(100 * v["replan_share_of_window"], "%.2f")))
    # rocprofv3 kernel stats: the ten largest kernels of the profiled inference command
    ks = os.path.join(P, "%s_infer608_b64_kernel_stats.csv" % out)
    if os.path.exists(ks):
        rows = sorted(csv.DictReader(open(ks)), key=lambda q: -float(q["TotalDurationNs"]))
        tot = sum(float(q["TotalDurationNs"]) for q in rows)
        L += ["", "## rocprofv3 --kernel-trace --stats of the inference command (`%s_infer608_b64_kernel_stats.csv`)" % out, "",
              "| kernel | calls | avg µs | share |", "|---|---|---|---|"]
        for q in rows[:8]:
            L.append("| `%s` | %s | %.1f | %.1f %% |" % (short(q["Name"])[:80], q["Calls"], float(q["AverageNs"]) / 1e3, 100 * float(q["TotalDurationNs"]) / tot))
    for wname, title in (("infer608_b64", "exact inference"), ("infer608_b64_split", "split inference"), ("train416_b16", "training")):
        mf, hb = _load("%s_%s_pmc_mfma.json" % (out, wname)), _load("%s_%s_pmc_hbm.json" % (out, wname))
        if not mf or not hb:
            continue
        L += ["", "## Counters, %s (`%s_%s_pmc_mfma.json`, `_pmc_hbm.json`; separate passes)" % (title, out, wname), "",
              "| kernel | launches | MFMA-busy | CU-busy | VALU : MFMA instructions | HBM MB per launch |", "|---|---|---|---|---|---|"]
        ks_ = sorted(mf["kernels"].items(), key=lambda kv: -kv[1]["dispatches"] * kv[1]["mfma_busy_frac"])[:8]
        for k, v in ks_:
            ins = v["insts_per_launch"]
            ratio = ins.get("valu", 0) / ins["mfma"] if ins.get("mfma") else None
            h = hb["kernels"].get(k, {})
            L.append("| `%s` | %d | %.3f | %.3f | %s | %s |" % (k[:70], v["dispatches"], v["mfma_busy_frac"], v["cu_busy_frac"], _f(ratio, "%.2f"), _f(h.get("hbm_MB_per_launch"))))
    open(os.path.join(P, "%s_summary.md" % out), "w").write("\n".join(L) + "\n")
    print("wrote profiles/%s_summary.md" % out)


write_summary()
