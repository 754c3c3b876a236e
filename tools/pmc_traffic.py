"""HBM traffic of the GEMM-family kernels from two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE; separate runs, as
/opt/skills/guides/MI355X_MICROARCH.md prescribes: the two do not fit one pass).

    python tools/pmc_traffic.py <fetch.db> <write.db> --steps 5 --bench-json gpurun_out/rNN_bench.json [--json profiles/rNN_pmc_traffic.json]

--bench-json: the JSON line of the same build's `python bench.py` run; its build_id (hash of the sources), its
launches_per_step (one launch = one rt_conv_gemm / rt_conv_wgrad(_grouped) call; a call may dispatch several kernels, e.g.
tile kernel + split reduction, so kernel_dispatches_per_step below is larger) and its workload are written into the output,
and bench.py only quotes a traffic file whose three values match the running build.

Units / corrections (same guide, "HBM [CDNA4]"): both counters are kilobytes; on gfx950 FETCH_SIZE reports half of the
bytes of wide coalesced reads (128-B requests tallied at 64 B), so fetch bytes = 2 x FETCH_SIZE x 1024; WRITE_SIZE
calibrates 1:1 on this run's own 607 MB gradient-buffer fill (592832 KB reported)."""
import argparse, json, re, sqlite3

FAMILY = re.compile(r"conv_gemm|conv_wgrad|wgrad_reduce|skinny_gemm|small_m_wgrad|w2_grouped|w2_reduce|bottleneck_")


def per_kernel(dbpath):
    db = sqlite3.connect(dbpath)
    out = {}
    for name, val in db.execute("select kernel_name, value from counters_collection"):
        a = out.setdefault(name, [0, 0.0]); a[0] += 1; a[1] += float(val)
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("fetch_db"); ap.add_argument("write_db")
    ap.add_argument("--steps", type=int, required=True, help="training steps the profiled command ran (warm-up included)")
    ap.add_argument("--json", default="")
    ap.add_argument("--bench-json", default="", help="bench.py's JSON line of the same build (build_id, launches_per_step)")
    ap.add_argument("--stats-md", default="", help="tools/rocpd_stats.py table of the same build's `rocprofv3 --kernel-trace --stats -- python bench.py` run: "
                    "the family's kernel-only time per step goes into the output (bench.py: roofline.kernel_only_frac)")
    a = ap.parse_args()
    f, w = per_kernel(a.fetch_db), per_kernel(a.write_db)
    rows, fam = [], dict(launches=0, fetch=0.0, write=0.0)
    for k in sorted(set(f) | set(w), key=lambda k: -(2 * f.get(k, [0, 0])[1] + w.get(k, [0, 0])[1])):
        n = f.get(k, w.get(k))[0]
        fb, wb = 2.0 * f.get(k, [0, 0.0])[1] * 1024, w.get(k, [0, 0.0])[1] * 1024
        rows.append((k, n, fb, wb))
        if FAMILY.search(k):
            fam["launches"] += n; fam["fetch"] += fb; fam["write"] += wb
    print("%-90s %7s %12s %12s" % ("kernel", "calls", "fetch MB/call", "write MB/call"))
    for k, n, fb, wb in rows[:25]:
        print("%-90s %7d %12.2f %12.2f" % (k[:90], n, fb / n / 1e6, wb / n / 1e6))
    # model construction (parameter init: ~500 torch fill / RNG kernels, once per process) is not part of a step
    init = re.compile(r"FillFunctor|distribution_|uniform_|normal_")
    step_rows = [r for r in rows if not init.search(r[0])]
    tot_f, tot_w = sum(r[2] for r in step_rows), sum(r[3] for r in step_rows)
    res = {
        "steps": a.steps,
        "gemm_family": {"kernel_dispatches_per_step": fam["launches"] / a.steps,
                        "hbm_bytes_per_step": (fam["fetch"] + fam["write"]) / a.steps,
                        "fetch_bytes_per_step": fam["fetch"] / a.steps, "write_bytes_per_step": fam["write"] / a.steps},
        "whole_step": {"hbm_bytes_per_step": (tot_f + tot_w) / a.steps, "fetch_bytes_per_step": tot_f / a.steps,
                       "write_bytes_per_step": tot_w / a.steps},
        "excluded_init_bytes": sum(r[2] + r[3] for r in rows if init.search(r[0])),
        "corrections": "fetch = 2 x FETCH_SIZE KB (gfx950 128-B requests tallied at 64 B); write = WRITE_SIZE KB",
    }
    if a.stats_md:
        # | kernel | calls | total ms | ... ; a step = one decoder_fwd_kernel dispatch (one cooperative launch per forward)
        fam_ms, steps = 0.0, 0
        for line in open(a.stats_md):
            c = [x.strip() for x in line.split("|")]
            if len(c) < 5 or not c[2].isdigit():
                continue
            if c[1].startswith("decoder_fwd_kernel"):
                steps = int(c[2])
            if FAMILY.search(c[1]):
                fam_ms += float(c[3])
        if steps:
            res["gemm_family"]["kernel_only_ms_per_step"] = fam_ms / steps
            res["gemm_family"]["kernel_only_source"] = "rocprofv3 --kernel-trace --stats of `python bench.py` (graph replay), sum of the family's kernel durations / steps"
    if a.bench_json:
        import os
        b = json.loads(open(a.bench_json).read().strip().splitlines()[-1])
        res["build_id"] = b.get("build_id")
        res["bench_launches_per_step"] = b["roofline"]["launches_per_step"]
        res["workload"] = [b["config"]["batch_per_gpu"], b["config"]["image_size"], b["n_gpus"]]
        res["gemm_family"]["hbm_bytes_per_launch"] = res["gemm_family"]["hbm_bytes_per_step"] / res["bench_launches_per_step"]
    print(json.dumps(res, indent=1))
    if a.json:
        json.dump(res, open(a.json, "w"), indent=1)


if __name__ == "__main__":
    main()
