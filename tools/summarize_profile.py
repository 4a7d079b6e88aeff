#!/usr/bin/env python3
"""Condense the rocprofv3 output of tools/profile_all.sh into profiles/: per case a text summary (kernel durations from
--kernel-trace, FETCH_SIZE / WRITE_SIZE per kernel from the separate --pmc passes) and the entries of
profiles/traffic.json that bench.py reports as roofline.traffic (keyed by workload AND by the hash of the code they were
measured on).

gfx950 counter conventions (MI355X_MICROARCH.md, HBM): rocprofv3 reports both counters in KiB; FETCH_SIZE tallies the
128-byte requests of a wide coalesced stream at 64 B, i.e. HALF the bytes -> x2; WRITE_SIZE is exact for 16-byte-per-lane
streaming stores and atomics.  Both corrections were re-checked on this access pattern in the same setup
(stream_read_kernel: 8 GiB read -> 4.0 GB reported; generate_kernel: 8 GB written -> 8.0 GB reported)."""
import csv
import glob
import json
import os
import re
import sys

NOT_A_STEP = re.compile(r"generate_kernel|stream_read|__amd_rocclr")


def find(root, pattern):
    """the files of the NEWEST run only: gpurun merges every run's output into the same scratch directories, and the files
    of an earlier run of the same case (another process id in the name) must not be counted on top"""
    files = sorted(glob.glob(os.path.join(root, "**", pattern), recursive=True))
    if len(files) <= 1:
        return files
    newest = max(os.path.getmtime(f) for f in files)
    return [f for f in files if newest - os.path.getmtime(f) < 120.0]


def bench_line(path):
    try:
        for line in open(path):
            line = line.strip()
            if line.startswith("{") and '"metric"' in line:
                return json.loads(line)
    except OSError:
        pass
    return None


def short(name):
    name = re.sub(r"\(.*", "", name)
    return name.replace("qe::pn::", "").replace("qe::", "").replace("void ", "")


def kernel_durations(case_dir):
    per = {}
    for f in find(os.path.join(case_dir, "trace"), "*kernel_trace.csv"):
        for r in csv.DictReader(open(f)):
            k = short(r.get("Kernel_Name", ""))
            d = int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
            e = per.setdefault(k, {"n": 0, "tot": 0, "min": 1 << 62, "max": 0, "vgpr": r.get("VGPR_Count"), "sgpr": r.get("SGPR_Count"),
                                   "lds": r.get("LDS_Block_Size"), "wg": r.get("Workgroup_Size")})
            e["n"] += 1; e["tot"] += d; e["min"] = min(e["min"], d); e["max"] = max(e["max"], d)
    return per


def counter_per_kernel(case_dir, sub, counter):
    """per kernel: the counter values of its dispatches, in dispatch order"""
    per = {}
    for f in find(os.path.join(case_dir, sub), "*counter_collection.csv"):
        rows = [r for r in csv.DictReader(open(f)) if r.get("Counter_Name") == counter]
        rows.sort(key=lambda r: int(r.get("Dispatch_Id", 0)))
        for r in rows:
            per.setdefault(short(r.get("Kernel_Name", "")), []).append(float(r["Counter_Value"]) * 1024.0)   # KiB -> bytes
    return per


def timed_region(values, warmup, steps):
    """The dispatches of one kernel that belong to the K timed steps: a kernel launched c times per step ran c * (W + K)
    times -- keep the last c * K; a kernel that ran fewer than K times (an exploring / first-execution form) is not part of
    the timed region."""
    n = len(values)
    if n >= warmup + steps and n % (warmup + steps) == 0:
        return values[-(n // (warmup + steps)) * steps:]
    if n >= steps:
        return values[-steps:]
    return []


def main(out, tag):
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    prof = os.path.join(root, "profiles")
    tpath = os.path.join(prof, "traffic.json")
    try:
        traffic = json.load(open(tpath))
        if "entries" not in traffic:
            traffic = {"entries": {}}
    except Exception:
        traffic = {"entries": {}}
    lines_all = []
    for case_dir in sorted(glob.glob(os.path.join(out, "*"))):
        if not os.path.isdir(case_dir):
            continue
        case = os.path.basename(case_dir)
        L = [f"# {tag} / {case}: rocprofv3 summary (tools/profile_all.sh; raw output under gpurun_out/, not tracked)"]
        plain = bench_line(os.path.join(case_dir, "plain.out"))
        if plain:
            r = plain["roofline"]
            L.append(f"un-profiled: {plain['config']['workload']}")
            L.append(f"  rows {plain['config']['rows_per_gpu']}  selected {plain['config']['selected_rows_total']}  ms/step {plain['ms_per_step']:.3f} "
                     f"(median {plain.get('ms_median', 0):.3f}, min {plain.get('ms_min', 0):.3f})  kernel {r['kernel_ms']:.3f} ms "
                     f"(median {r.get('kernel_ms_median', 0):.3f}, min {r.get('kernel_ms_min', 0):.3f})  e2e+D2H {plain.get('e2e_with_d2h_ms') or 0:.1f} ms")
            L.append(f"  algorithmic {r['algorithmic_bytes_per_launch'] / 1e9:.2f} GB -> {r['achieved']:.0f} GB/s = {r['frac']:.3f} of 8 TB/s; rows/s {plain['value']:.3e}")
        else:
            for line in open(os.path.join(case_dir, "plain.out")) if os.path.exists(os.path.join(case_dir, "plain.out")) else []:
                L.append("un-profiled: " + line.rstrip())
        dur = kernel_durations(case_dir)
        if dur:
            L.append("kernel-trace pass (durations under the profiler; a few % above un-profiled: DVFS, MI355X_MICROARCH.md):")
            for k, e in sorted(dur.items(), key=lambda kv: -kv[1]["tot"])[:14]:
                L.append(f"  {k[:60]:60s} calls {e['n']:4d}  avg {e['tot'] / e['n'] / 1e6:8.4f} ms  min {e['min'] / 1e6:8.4f}  max {e['max'] / 1e6:8.4f}  "
                         f"vgpr {e['vgpr']} sgpr {e['sgpr']} lds {e['lds']} wg {e['wg']}")
        fetch = counter_per_kernel(case_dir, "pmc_fetch", "FETCH_SIZE")
        write = counter_per_kernel(case_dir, "pmc_write", "WRITE_SIZE")
        pj = bench_line(os.path.join(case_dir, "pmc_fetch.out"))
        W_, K_ = (pj["warmup"], pj["steps"]) if pj else (0, 1)
        step_fetch = step_write = 0.0
        if fetch or write:
            L.append("PMC passes (separate runs), per kernel, average per dispatch of the timed steps (all dispatches in brackets): "
                     "FETCH_SIZE raw | x2 (gfx950 correction) | WRITE_SIZE:")
            def avg(v):
                return sum(v) / len(v) if v else 0.0
            for k in sorted(set(fetch) | set(write), key=lambda k: -(2 * sum(fetch.get(k, [])) + sum(write.get(k, [])))):
                f_all, w_all = fetch.get(k, []), write.get(k, [])
                f_t, w_t = (timed_region(f_all, W_, K_), timed_region(w_all, W_, K_)) if pj else (f_all, w_all)
                in_step = not NOT_A_STEP.search(k) and (f_t or w_t)
                L.append(f"  {k[:60]:60s} n {len(f_t):3d} [{len(f_all):3d}]  fetch {avg(f_t) / 1e9:8.3f} GB | {2 * avg(f_t) / 1e9:8.3f} GB | write {avg(w_t) / 1e9:8.3f} GB"
                         + ("" if in_step else "   (outside the timed steps)"))
                if in_step:
                    step_fetch += sum(f_t)
                    step_write += sum(w_t)
        if not pj and dur:
            # a tool run (no bench line): several executions, maybe of several forms, share kernel names -- list the dispatches
            # in order, each with its own duration and its own counters (the n-th dispatch of a kernel in the trace pass is
            # the n-th in the PMC passes: the same command)
            rows = []
            for f in find(os.path.join(case_dir, "trace"), "*kernel_trace.csv"):
                rows += list(csv.DictReader(open(f)))
            rows.sort(key=lambda r: int(r["Start_Timestamp"]))
            seen = {}
            L.append("dispatch by dispatch (kernel-trace pass; FETCH_SIZE x2 and WRITE_SIZE of the same dispatch from the PMC passes):")
            for r in rows:
                k = short(r.get("Kernel_Name", ""))
                i = seen.get(k, 0)
                seen[k] = i + 1
                if NOT_A_STEP.search(k) or k.startswith("__amd"):
                    continue
                f_, w_ = fetch.get(k, []), write.get(k, [])
                d = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6
                fb = 2 * f_[i] / 1e9 if i < len(f_) else float("nan")
                wb = w_[i] / 1e9 if i < len(w_) else float("nan")
                if d < 0.02 and fb + wb < 0.05:
                    continue
                L.append(f"  {k[:28]:28s} {d:8.3f} ms  grid {r.get('Grid_Size', r.get('Grid_Size_X', '?')):>8s} wg {r.get('Workgroup_Size', r.get('Workgroup_Size_X', '?')):>4s} "
                         f"lds {r.get('LDS_Block_Size', '?'):>6s}  read {fb:7.2f} GB  written {wb:7.2f} GB  -> {(fb + wb) / d:6.2f} TB/s" if d > 0 else "")
        if pj and (step_fetch or step_write):
            hbm = (2 * step_fetch + step_write) / K_
            r = pj["roofline"]
            kms = (plain or pj)["roofline"]["kernel_ms"]
            L.append(f"HBM traffic of the timed step's kernels per launch ({K_} steps): fetch x2 {2 * step_fetch / K_ / 1e9:.3f} GB + write "
                     f"{step_write / K_ / 1e9:.3f} GB = {hbm / 1e9:.3f} GB (algorithmic {r['algorithmic_bytes_per_launch'] / 1e9:.3f} GB)")
            L.append(f"  over the un-profiled kernel time {kms:.3f} ms: {hbm / kms / 1e6:.0f} GB/s = frac_moved {hbm / kms / 1e6 / 8000:.3f} of 8 TB/s "
                     f"(frac on algorithmic bytes {r['algorithmic_bytes_per_launch'] / kms / 1e6 / 8000:.3f})")
            traffic["entries"][r["traffic_key"]] = {
                "hbm_bytes_per_launch": hbm, "fetch_bytes_raw": step_fetch / K_, "fetch_bytes_x2": 2 * step_fetch / K_,
                "write_bytes": step_write / K_, "code_hash": r["code_hash"], "launches": K_,
                "source": f"profiles/{tag}_{case}_summary.txt (rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE, separate passes)"}
        text = "\n".join(L) + "\n"
        open(os.path.join(prof, f"{tag}_{case}_summary.txt"), "w").write(text)
        lines_all.append(text)
    traffic["note"] = ("HBM bytes per launch of the kernels inside the timed region: FETCH_SIZE x2 (gfx950 tallies 128-B requests at 64 B) + "
                       "WRITE_SIZE, rocprofv3 KiB units; bench.py reports an entry only when its code_hash matches the running code")
    json.dump(traffic, open(tpath, "w"), indent=1, sort_keys=True)
    print("\n".join(lines_all))


if __name__ == "__main__":
    main(sys.argv[1] if len(sys.argv) > 1 else "gpurun_out/prof_r02", sys.argv[2] if len(sys.argv) > 2 else "r02")
