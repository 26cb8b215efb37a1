"""Per-layer HBM traffic of one inference step: joins the per-dispatch FETCH_SIZE / WRITE_SIZE counters of two
`rocprofv3 --pmc` runs of `bench.py --steps 1 --warmup 1` (tools/profile_round.sh) with the per-layer table of
tools/layer_profile.py by launch order, and prints measured vs algorithmic bytes per conv launch.

    python tools/layer_traffic.py gpurun_out/r02_inf_fetch gpurun_out/r02_inf_write profiles/r02_layers_608_b64.txt
"""
import csv
import glob
import sys


def per_dispatch(d, counter):
    f = sorted(glob.glob(d + "/**/*counter_collection.csv", recursive=True))[-1]
    rows = {}
    for r in csv.DictReader(open(f)):
        if r["Counter_Name"] != counter:
            continue
        rows[int(r["Dispatch_Id"])] = (r["Kernel_Name"], float(r["Counter_Value"]))
    return [rows[k] for k in sorted(rows)]


def main(fetch_dir, write_dir, layers_txt):
    fe = per_dispatch(fetch_dir, "FETCH_SIZE")
    wr = per_dispatch(write_dir, "WRITE_SIZE")
    names = []
    for line in open(layers_txt):
        p = line.split()
        if len(p) >= 6 and ("|" in p[0] or p[0].startswith("stages.0.0")):
            names.append((p[0], float(p[2]), float(p[3]), float(p[5])))  # ms, GFLOP, GB/s (algorithmic)
    is_conv = lambda n: "conv_igemm_kernel" in n or "stem_kernel" in n
    fconv = [x for x in fe if is_conv(x[0])]
    wconv = [x for x in wr if is_conv(x[0])]
    n = len(names)
    # the last step's launches (warm-up + timed step are both in the file)
    fconv, wconv = fconv[-n:], wconv[-n:]
    print("%-36s %9s %9s %9s %7s" % ("launch", "alg MB", "fetch MB", "write MB", "x alg"))
    tot_a = tot_m = 0.0
    for (nm, ms, gf, gbs), f, w in zip(names, fconv, wconv):
        alg = gbs * ms * 1e-3 * 1e3  # GB/s * ms -> MB
        fb = f[1] * 2 * 1024 / 1e6   # FETCH_SIZE in KiB; gfx950: x2 (MI355X_MICROARCH.md)
        wb = w[1] * 1024 / 1e6
        tot_a += alg
        tot_m += fb + wb
        print("%-36s %9.1f %9.1f %9.1f %7.2f" % (nm, alg, fb, wb, (fb + wb) / alg if alg else 0))
    print("%-36s %9.1f %19.1f %7.2f" % ("TOTAL convs", tot_a, tot_m, tot_m / tot_a))


if __name__ == "__main__":
    main(*sys.argv[1:4])
