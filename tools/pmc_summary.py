"""Aggregates rocprofv3 PMC passes of ONE bench command (each pass = `--kernel-trace --pmc <counters>` only, program
directly after `--`) into per-kernel averages -> profiles/<name>.json.

  python tools/pmc_summary.py profiles/r2_pmc_summary.json gpurun_out/r2_pmc_sq gpurun_out/r2_pmc_fetch gpurun_out/r2_pmc_write

Units and the gfx950 corrections follow /opt/skills/guides/MI355X_MICROARCH.md:
  * FETCH_SIZE / WRITE_SIZE are KiB; FETCH_SIZE reports half the bytes of wide coalesced reads on gfx950 -> doubled;
    Infinity-Cache hits are counted (fabric-side counters).
  * GRBM_GUI_ACTIVE is summed over the 8 XCDs: effective clock = GRBM_GUI_ACTIVE / 8 / kernel duration.
  * SQ_VALU_MFMA_BUSY_CYCLES counts cycles per SIMD summed over the chip: MFMA-pipe utilisation =
    SQ_VALU_MFMA_BUSY_CYCLES / (4 SIMDs x 256 CUs x GRBM_GUI_ACTIVE / 8).
  * SQ_WAVE_CYCLES / SQ_BUSY_CYCLES count quad-cycles.
Durations come from the Start/End timestamps of the same (profiled) dispatches: they are a few % longer than in an
un-profiled run (the guide: never compare a profiled arm with an un-profiled one)."""
import csv
import glob
import json
import os
import re
import sys


def canonical(name: str):
    """rocprofv3 prints some instantiations mangled and some through a lossy demangler: map both to bench.py's names."""
    if "cir" not in name:
        return None
    m8 = re.search(r"gemm256_kernelIDF16_Lb1ELb([01])EfLi(n?)(\d)ELb0ELb1ELb([01])E", name)
    if m8:                                                # "split8" operands (round 6): MIX instantiations, named as ops._gemm_split8 records them
        return "cir::gemm256_kernel<split8%s>" % (",split8-out" if m8.group(4) == "1" else (",residual" if m8.group(1) == "1" else ""))
    if "gemm_split8_kernel" in name:
        return "cir::gemm_split8_kernel"
    for short in ("layernorm_split8_kernel", "split8_kernel", "split16_kernel"):
        if short in name:
            return f"cir::{short}"
    m = re.search(r"gemm256_kernelI(DF16b|DF16_|Dh)Lb([01])ELb([01])E(f|DF16_|Dh)Li(n?)(\d)E(?:Lb([01])E)?", name)
    if m:                                                 # names as ops.gemm_kernel_name prints them (ACT: template constant or run time)
        t = "__bf16" if m.group(1) == "DF16b" else "_Float16"
        f32, res = m.group(2) == "1", "true" if m.group(3) == "1" else "false"
        stream16 = m.group(4) in ("DF16_", "Dh")
        act = None if m.group(5) == "n" else int(m.group(6))
        base = f"cir::gemm256_kernel<{t},{'true' if f32 else 'false'},{res}"
        if stream16:
            return base + (",_Float16,0>" if act == 0 else ",_Float16>")
        if not f32 and act in (0, 1):
            return base + f",float,{act}" + (",true>" if m.group(7) == "1" else ">")   # ",true" = LayerNorm folded in (cir_gemm_ln_bias_act)
        return base + ">"
    if "gemm256_kernel<" in name:                         # lossy demangle: keeps the trailing template arguments only
        st = ",_Float16" if "_Float16" in name else ""
        res = "true" if re.search(r"true(, *[_A-Za-z0-9-]+)*>", name) else "false"
        return f"cir::gemm256_kernel<?,true,{res}{st}>"
    if "gemm_kernelIfLi1E" in name or "gemm_kernel<float" in name:
        return "cir::gemm_kernel<float,1>"
    m = re.search(r"gemm_kernelI(DF16b|DF16_|Dh)Li([012])E", name)
    if m:
        return f"cir::gemm_kernel<{'__bf16' if m.group(1) == 'DF16b' else '_Float16'},{m.group(2)}>"
    if "gemm_kernel<" in name:
        return "cir::gemm_kernel<?>"
    for short in ("xattn_fold_kernel", "attn_f32_kernel"):
        if short in name:
            return f"cir::{short}"
    for short in ("attn_shared_kernel", "attn_stream_kernel", "layernorm_h16_kernel", "layernorm_kernel", "embed_ln_kernel", "patchify_kernel", "vit_assemble_kernel",
                  "gather_rows_kernel", "topk_desc_kernel", "small_linear_kernel", "cls_xattn_kernel"):
        if short in name:
            masked = ""
            if short.startswith("attn"):
                mm = re.search(short + r"I(DF16b|DF16_|Dh)Lb([01])E", name)
                masked = ("<masked>" if (mm and mm.group(2) == "1") or ("true>" in name and not mm) else "<unmasked>")
            return f"cir::{short}{masked}"
    for short in ("wgrad_kernel", "tattn_fwd_kernel", "tattn_bwd_dq_kernel", "tattn_bwd_dkv_kernel", "ln_bwd_fused_kernel", "res_ln_train_kernel",
                  "rows16_colsum_kernel", "rows_scale_add_kernel", "bmm_mfma_kernel", "eltwise_kernel", "adamw_kernel", "transpose_multi_kernel",
                  "layernorm_bwd_kernel", "colsum_kernel", "embed_bwd_kernel"):      # the training step's kernels (train*.hip)
        if short in name:
            mm = re.search(short + r"I(DF16b|DF16_|Dh)Lb([01])E", name) if short.startswith("tattn") else None
            return f"cir::{short}" + (("<masked>" if mm.group(2) == "1" else "<unmasked>") if mm else "")
    return "cir::other"


def main():
    out, dirs = sys.argv[1], sys.argv[2:]
    ker = {}
    passes = {}
    for d in dirs:
        files = glob.glob(f"{d}/**/*counter_collection.csv", recursive=True)
        seen = set()
        for f in files:
            for r in csv.DictReader(open(f)):
                k = canonical(r["Kernel_Name"])
                if k is None:
                    continue
                e = ker.setdefault(k, {"counters": {}, "dur_ns": {}, "n": {}})
                c = r["Counter_Name"]
                seen.add(c)
                e["counters"][c] = e["counters"].get(c, 0.0) + float(r["Counter_Value"])
                e["n"][c] = e["n"].get(c, 0) + 1
                e["dur_ns"][c] = e["dur_ns"].get(c, 0.0) + (int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
        passes[d] = sorted(seen)
    # what was profiled: taken from the bench line the SAME pass printed (never from the environment) + the kernel-source hash
    meta = {}
    for d in dirs:
        jf = d.rstrip("/") + ".json"
        if os.path.exists(jf):
            for ln in open(jf):
                if ln.startswith('{"metric"'):
                    cfg = json.loads(ln)
                    if "queries_per_step_per_gpu" in cfg["config"]:
                        meta = {"residual_stream": cfg["config"]["residual_stream"], "queries": cfg["config"]["queries_per_step_per_gpu"],
                                "k": cfg["config"]["k"], "subset": cfg["config"]["subset"], "image_size": cfg["config"]["image_size"], "dtype": cfg["dtype"]}
                    else:                                  # `--mode train` (or another secondary mode): keep its own workload string
                        meta = {"workload": cfg["config"]["workload"], "dtype": cfg["dtype"]}
            break
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    import bench
    res = {**meta, "csrc_sha16": bench.csrc_sha16(),
           "source": "rocprofv3 --kernel-trace --pmc <counters> -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-precision-table "
                     "(one pass per counter group; all launches of the run incl. warm-up and the instrumented steps)",
           "passes": passes, "by_kernel": {}}
    for k, e in sorted(ker.items()):
        avg = {c: e["counters"][c] / e["n"][c] for c in e["counters"]}
        dur_us = {c: e["dur_ns"][c] / e["n"][c] / 1e3 for c in e["counters"]}
        row = {"launches_per_pass": max(e["n"].values()), "avg_us_profiled": round(sum(dur_us.values()) / len(dur_us), 2),
               "counters_per_launch": {c: round(v, 1) for c, v in avg.items()}}
        if "GRBM_GUI_ACTIVE" in avg:
            cyc = avg["GRBM_GUI_ACTIVE"] / 8.0
            # GRBM_GUI_ACTIVE spans more than the kernel on short dispatches (the guide: the quotient reads high below ~0.3 ms - 3.6 / 5.4 GHz
            # came out for 20-us kernels): no clock is quoted there, and the utilisation figures below are per GRBM cycle, not per kernel cycle
            row["effective_clock_ghz"] = round(cyc / (dur_us["GRBM_GUI_ACTIVE"] * 1e3), 3) if dur_us["GRBM_GUI_ACTIVE"] >= 300.0 else None
            if "SQ_VALU_MFMA_BUSY_CYCLES" in avg:
                row["mfma_pipe_utilisation"] = round(avg["SQ_VALU_MFMA_BUSY_CYCLES"] / (4 * 256 * cyc), 4)
            if "SQ_BUSY_CYCLES" in avg:
                row["sq_busy_quad_cycles_per_gpu_cycle"] = round(avg["SQ_BUSY_CYCLES"] / cyc, 3)
        if "FETCH_SIZE" in avg or "WRITE_SIZE" in avg:
            fb = 2.0 * avg.get("FETCH_SIZE", 0.0) * 1024.0
            wb = avg.get("WRITE_SIZE", 0.0) * 1024.0
            row["fetch_bytes_per_launch_corrected_x2"] = round(fb)
            row["write_bytes_per_launch"] = round(wb)
            row["hbm_bytes_per_launch"] = round(fb + wb)
            du = dur_us.get("FETCH_SIZE", dur_us.get("WRITE_SIZE"))
            row["hbm_side_tb_per_s"] = round((fb + wb) / (du * 1e-6) / 1e12, 3)
        res["by_kernel"][k] = row
    json.dump(res, open(out, "w"), indent=1)
    for k, row in res["by_kernel"].items():
        print(k, json.dumps({a: b for a, b in row.items() if a != "counters_per_launch"}))


if __name__ == "__main__":
    main()
