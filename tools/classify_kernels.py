"""rocprofv3 kernel_stats.csv -> GPU time by kernel family of the encoder / ingest path (JSON on stdout)."""
import csv, json, sys
fam = {"gemm": 0.0, "gelu / elementwise (torch)": 0.0, "attention (librdx)": 0.0, "attention (torch SDPA + scatter / gather / transposes)": 0.0,
       "add + LayerNorm (librdx)": 0.0, "LayerNorm / add (torch)": 0.0, "K1 normalise + index add (librdx)": 0.0, "encoder stage kernels (librdx)": 0.0,
       "copies / fills": 0.0, "other": 0.0}
rows = list(csv.DictReader(open(sys.argv[1])))
for r in rows:
    n, t = r["Name"], float(r["TotalDurationNs"]) / 1e6
    l = n.lower()
    if "k_enc_attention" in n or "k_enc_attn" in n: fam["attention (librdx)"] += t
    elif "k_enc_add_ln" in n or "k_enc_ln_rows" in n: fam["add + LayerNorm (librdx)"] += t
    elif "k_enc_stage" in n or "k_enc_linear" in n or "k_enc_embed" in n: fam["encoder stage kernels (librdx)"] += t
    elif "rdx::" in n or "_ZN3rdx" in n: fam["K1 normalise + index add (librdx)"] += t
    elif "cijk" in l or "gemm" in l or "hipblaslt" in l or l.startswith("mt") or "_mt" in l[:40]: fam["gemm"] += t
    elif "attention" in l or "fmha" in l or "flash" in l or "softmax" in l or "index_select" in l or "index_copy" in l or "indexfunc" in l or "gather" in l or "scatter" in l: fam["attention (torch SDPA + scatter / gather / transposes)"] += t
    elif "layer_norm" in l or "layernorm" in l: fam["LayerNorm / add (torch)"] += t
    elif "gelu" in l or "elementwise" in l: fam["gelu / elementwise (torch)"] += t
    elif "copy" in l or "fill" in l: fam["copies / fills"] += t
    else: fam["other"] += t
tot = sum(fam.values())
top = sorted(rows, key=lambda r: -float(r["TotalDurationNs"]))[:8]
print(json.dumps({"gpu_ms_total": round(tot, 2), "gpu_ms_by_family": {k: round(v, 2) for k, v in fam.items() if v > 0},
                  "share": {k: round(v / tot, 4) for k, v in fam.items() if v > 0},
                  "top_kernels": [{"name": r["Name"][:90], "calls": int(r["Calls"]), "ms": round(float(r["TotalDurationNs"]) / 1e6, 2)} for r in top]}))
