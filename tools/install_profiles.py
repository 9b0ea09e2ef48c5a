"""Copies the artifacts of tools/profile_all.sh (gpurun_out/final) into profiles/ (round-2 names) and prints the figures the docs quote."""
import collections, csv, glob, json, os, re, shutil, sys
O = sys.argv[1] if len(sys.argv) > 1 else "gpurun_out/final"
R = "profiles/r02_"
newest = lambda pat: max(glob.glob(pat), key=os.path.getmtime)
shutil.copy(newest(O + "/prof_b/*/*_kernel_stats.csv"), R + "bench_kernel_stats.csv")
shutil.copy(newest(O + "/prof_bs/*/*_kernel_stats.csv"), R + "bench_kernel_stats_sequential.csv")
shutil.copy(O + "/bench.json", R + "bench.json")
shutil.copy(O + "/layers.txt", R + "forward_layers_b256.txt")
shutil.copy(O + "/layers32.txt", R + "forward_f32_layers_b256.txt")
shutil.copy(newest(O + "/prof_f32/*/*_kernel_stats.csv"), R + "bench_f32_kernel_stats_sequential.csv")
shutil.copy(O + "/postproc.txt", R + "postproc.txt")
shutil.copy(O + "/merge_scaling.txt", R + "merge_scaling.txt")
open(R + "conv_phase_stamps.txt", "w").write("# tools/stamp_conv.sh: s_memtime stamps inside k_conv_igemm / k_conv3_pair / k_front (diagnostic build), one eager forward of 256 tiles, fp16;\n# cycles per wave and tile, phases as named on each line (k_conv_igemm: waiting at the top-of-stage barrier | LDS staging incl. the wait for the\n# prefetched global loads | issuing the next prefetch | the MFMA k loop (issue only) | the epilogue (bias, SiLU, stores))\n" + "".join(l for l in open(O + "/stamps.txt") if l.startswith("STAMPS")))
shutil.copy(newest(O + "/pptrace/*/*_kernel_stats.csv"), R + "decode_nms_kernel_stats.csv")
KER = ("obb::k_conv", "k_dwconv3", "k_maxpool5", "k_upsample2", "k_attention", "k_stem_conv", "k_front", "k_sppf_pools", "k_bneck_stripe", "k_c3k_image", "k_dwpw_stripe", "_f32")
def load(d):
    plan = [l for l in open(d + "/plan.txt").read().strip().split("\n") if not l.startswith("total")]
    disp = collections.OrderedDict()
    for r in csv.DictReader(open(newest(d + "/*/*_counter_collection.csv"))):
        disp.setdefault(int(r["Dispatch_Id"]), {"name": r["Kernel_Name"]})[r["Counter_Name"]] = float(r["Counter_Value"])
    ds = [v for k, v in sorted(disp.items()) if any(t in v["name"] for t in KER)]
    return plan, ds[-len(plan):]
plan, fe = load(O + "/pmc_fetch"); _, wr = load(O + "/pmc_write")
tr = tw = 0; lines = []
for o, a, b in zip(plan, fe, wr):
    r = a.get("FETCH_SIZE", 0) * 2 * 1024 / 1e6; w = b.get("WRITE_SIZE", 0) * 1024 / 1e6
    tr += r; tw += w
    lines.append("%-78s read_x2_MB %8.1f write_MB %8.1f" % (o[:78], r, w))
hdr = ["# rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes), one forward of B=256 tiles 416x416x3, fp16, MI355X (final round-2 build)",
       "# FETCH_SIZE is doubled (gfx950 reports half the bytes of wide coalesced reads, MI355X_MICROARCH.md section HBM); counters are in KiB",
       "# forward total: read %.2f GB (x2 corrected), write %.2f GB -> %.1f MB / tile" % (tr / 1e3, tw / 1e3, (tr + tw) / 256)]
open(R + "forward_hbm_traffic_b256.txt", "w").write("\n".join(hdr + lines) + "\n")
print(hdr[2])
for src, dst in (("/pmc_sq", "forward_pmc_sq_b256.txt"), ("/pmc_sq32", "forward_f32_pmc_sq_b256.txt")):
    plan, sq = load(O + src)
    names = [c for c in sq[0] if c != "name"]
    with open(R + dst, "w") as f:
        f.write("counters: %s   (SQ_* in quad-cycles summed over waves; SQ_VALU_MFMA_BUSY_CYCLES in cycles summed over SIMDs)\n" % names)
        for o, v in zip(plan, sq):
            mf = v.get("SQ_VALU_MFMA_BUSY_CYCLES", 0) / max(1.0, 4.0 * v.get("SQ_WAVE_CYCLES", 1))
            f.write(o[:70].ljust(70) + " " + " ".join("%s=%.3g" % (c, v.get(c, 0)) for c in names) + "  mfma_busy/wave_cycles=%.3f\n" % mf)
for f, steps in ((R + "bench_kernel_stats.csv", 23), (R + "bench_kernel_stats_sequential.csv", 23)):
    rows = list(csv.DictReader(open(f)))
    fw = [r for r in rows if any(k.replace("obb::", "") in r["Name"] for k in KER)]
    t = sum(float(r["TotalDurationNs"]) for r in fw) / steps / 1e6; c = sum(int(r["Calls"]) for r in fw) / steps
    conv = [r for r in fw if "k_conv_igemm" in r["Name"]]
    ct = sum(float(r["TotalDurationNs"]) for r in conv) / steps / 1e6; cc = sum(int(r["Calls"]) for r in conv) / steps
    print(f, "forward kernels %.3f ms/step over %.0f launches; k_conv_igemm %.3f over %.0f (avg %.1f us)" % (t, c, ct, cc, ct / cc * 1e3))
    for r in sorted(rows, key=lambda r: -float(r["TotalDurationNs"]))[:12]:
        print("   %-72s %.3f ms/step  %d calls" % (r["Name"][:72], float(r["TotalDurationNs"]) / steps / 1e6, int(r["Calls"]) // steps))
for n in ("bench.json", "bench_prof_b.log", "bench_prof_bs.log", "bench_seq_plain.json", "bench_nopipe.json", "bench_prof_f32.log"):
    for l in open(O + "/" + n):
        if l.startswith("{"):
            d = json.loads(l); print(n, round(d["value"]), round(d["ms_per_step"], 3), "fwd", round(d["roofline"]["forward_ms"], 3), "TF", round(d["roofline"]["achieved"], 1), d.get("cpu_baseline", {}).get("value"), d.get("cpu_baseline", {}).get("sample"))
print(open(O + "/layers.txt").read().strip().split("\n")[-1])
print(open(O + "/layers32.txt").read().strip().split("\n")[-1])
print(open(O + "/postproc.txt").read())
print(open(O + "/merge_scaling.txt").read())
