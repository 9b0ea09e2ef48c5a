"""Copies the artifacts of tools/profile_all.sh (gpurun_out/final) into profiles/ (names prefixed with the round: r04_ by default, argv[2]) and prints the figures the docs quote.
Parts that were not collected (a PART of profile_all.sh not run) are skipped."""
import collections, csv, glob, json, os, shutil, sys
O = sys.argv[1] if len(sys.argv) > 1 else "gpurun_out/final"
ROUND = sys.argv[2] if len(sys.argv) > 2 else "r04"
R = "profiles/" + ROUND + "_"
newest = lambda pat: max(glob.glob(pat), key=os.path.getmtime)
KER = ("obb::k_conv", "k_dwconv3", "k_maxpool5", "k_upsample2", "k_attention", "k_stem_conv", "k_front", "k_sppf_pools", "k_bneck_stripe", "k_c3k_image", "k_dwpw_stripe", "_f32")


def cp(src, dst):
    try:
        shutil.copy(newest(src) if "*" in src else src, R + dst)
        return True
    except (ValueError, OSError):
        print("skipped (not collected):", src)
        return False


cp(O + "/bench.json", "bench.json")
cp(O + "/prof_f32/*/*_kernel_stats.csv", "bench_f32_kernel_stats_sequential.csv")
cp(O + "/prof_f16/*/*_kernel_stats.csv", "bench_f16_kernel_stats_sequential.csv")
cp(O + "/layers32.txt", "forward_f32_layers_b256.txt")
cp(O + "/layers32_512.txt", "forward_f32_layers_b512.txt")
cp(O + "/layers.txt", "forward_f16_layers_b256.txt")
cp(O + "/layers32_128.txt", "forward_f32_layers_128px_b8192.txt")
cp(O + "/layers_128.txt", "forward_f16_layers_128px_b8192.txt")
cp(O + "/postproc.txt", "postproc.txt")
cp(O + "/merge_scaling.txt", "merge_scaling.txt")
cp(O + "/pptrace/*/*_kernel_stats.csv", "decode_nms_kernel_stats.csv")


def load(d):
    plan = [l for l in open(d + "/plan.txt").read().strip().split("\n") if not l.startswith("total")]
    disp = collections.OrderedDict()
    for r in csv.DictReader(open(newest(d + "/*/*_counter_collection.csv"))):
        disp.setdefault(int(r["Dispatch_Id"]), {"name": r["Kernel_Name"]})[r["Counter_Name"]] = float(r["Counter_Value"])
    ds = [v for k, v in sorted(disp.items()) if any(t in v["name"] for t in KER)]
    return plan, ds[-len(plan):]


for tag, fdir, wdir, dst in (("fp32", "/pmc_fetch32", "/pmc_write32", "forward_f32_hbm_traffic_b256.txt"), ("fp16", "/pmc_fetch", "/pmc_write", "forward_hbm_traffic_b256.txt")):
    try:
        plan, fe = load(O + fdir)
        _, wr = load(O + wdir)
    except (ValueError, OSError):
        print("skipped (not collected):", fdir, wdir)
        continue
    tr = tw = 0
    lines = []
    for o, a, b in zip(plan, fe, wr):
        r = a.get("FETCH_SIZE", 0) * 2 * 1024 / 1e6
        w = b.get("WRITE_SIZE", 0) * 1024 / 1e6
        tr += r
        tw += w
        lines.append("%-78s read_x2_MB %8.1f write_MB %8.1f" % (o[:78], r, w))
    hdr = ["# rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes), one forward of B=256 tiles 416x416x3, %s arithmetic, MI355X (%s build)" % (tag, ROUND),
           "# FETCH_SIZE is doubled (gfx950 reports half the bytes of wide coalesced reads, MI355X_MICROARCH.md section HBM); counters are in KiB",
           "# forward total: read %.2f GB (x2 corrected), write %.2f GB -> %.1f MB / tile" % (tr / 1e3, tw / 1e3, (tr + tw) / 256)]
    open(R + dst, "w").write("\n".join(hdr + lines) + "\n")
    print(tag, hdr[2])

for src, dst in (("/pmc_sq32", "forward_f32_pmc_sq_b256.txt"), ("/pmc_sq", "forward_f16_pmc_sq_b256.txt")):
    try:
        plan, sq = load(O + src)
    except (ValueError, OSError):
        print("skipped (not collected):", src)
        continue
    names = [c for c in sq[0] if c != "name"]
    with open(R + dst, "w") as f:
        f.write("counters: %s   (SQ_* in quad-cycles summed over waves; SQ_VALU_MFMA_BUSY_CYCLES in cycles summed over SIMDs)\n" % names)
        for o, v in zip(plan, sq):
            mf = v.get("SQ_VALU_MFMA_BUSY_CYCLES", 0) / max(1.0, 4.0 * v.get("SQ_WAVE_CYCLES", 1))
            f.write(o[:70].ljust(70) + " " + " ".join("%s=%.3g" % (c, v.get(c, 0)) for c in names) + "  mfma_busy/wave_cycles=%.3f\n" % mf)

for f in (R + "bench_f32_kernel_stats_sequential.csv", R + "bench_f16_kernel_stats_sequential.csv"):
    if not os.path.exists(f):
        continue
    rows = list(csv.DictReader(open(f)))
    fw = [r for r in rows if any(k.replace("obb::", "") in r["Name"] for k in KER)]
    glue = [r for r in rows if "at::native" in r["Name"] or "rocprim" in r["Name"] or "at_cuda_detail" in r["Name"]]
    print(f, "forward kernels: total %.3f ms over %d launches (all steps incl. warm-up); torch / rocprim kernels in the trace: %d launches" %
          (sum(float(r["TotalDurationNs"]) for r in fw) / 1e6, sum(int(r["Calls"]) for r in fw), sum(int(r["Calls"]) for r in glue)))
    for r in sorted(rows, key=lambda r: -float(r["TotalDurationNs"]))[:14]:
        print("   %-90s %.3f ms total  %d calls" % (r["Name"][:90], float(r["TotalDurationNs"]) / 1e6, int(r["Calls"])))
    for r in glue[:12]:
        print("   GLUE %-84s %d calls" % (r["Name"][:84], int(r["Calls"])))
for n in ("bench.json", "bench_prof_f32.log", "bench_prof_f16.log", "bench_nopipe.json"):
    try:
        for l in open(O + "/" + n):
            if l.startswith("{"):
                d = json.loads(l)
                print(n, d["dtype"], round(d["value"]), "tiles/s", round(d["ms_per_step"], 3), "ms/step fwd", round(d["roofline"]["forward_ms"], 3), "TF", round(d["roofline"]["achieved"], 1),
                      "frac", round(d["roofline"]["frac"], 4), "| cpu", d.get("cpu_baseline", {}).get("value"))
                for k, v in d.get("also", {}).items():
                    if k == "f16":
                        print("   also.f16", round(v["value"]), "tiles/s fwd", round(v["roofline"]["forward_ms"], 3), "frac", round(v["roofline"]["frac"], 4),
                              "| dual", round(v["dual_scale"]["ms_per_step"], 2), "ms frac", round(v["dual_scale"]["roofline"]["frac"], 4), v["dual_scale"]["forward_ms"])
                    else:
                        print("   also." + k, round(v["value"]), round(v["ms_per_step"], 2), "ms frac", round(v["roofline"]["frac"], 4), v.get("forward_ms"))
    except OSError:
        pass
for n in ("layers32.txt", "layers32_512.txt", "layers.txt", "layers32_128.txt", "layers_128.txt"):
    try:
        print(n, open(O + "/" + n).read().strip().split("\n")[-1])
    except OSError:
        pass
