"""One paragraph for profiles/README.md from an installed sequence:   python tools/profile_summary.py <tag>"""
import json
import re
import sys
from pathlib import Path

P = Path(__file__).resolve().parent.parent / "profiles"
tag = sys.argv[1]
d = json.load(open(P / f"{tag}_bench.json"))
s, e = d["search"], d["extra"]
k = lambda v: f"{v / 1e3:.1f} k"


def kern(pattern, path):
    for line in open(path):
        if re.search(pattern, line):
            m = re.search(r"avg\s+([\d.]+) us", line)
            return float(m.group(1))
    return float("nan")


rk = P / f"{tag}_roofline_kernel_stats.txt"
ck = P / f"{tag}_cnn14_kernel_stats.txt"
c = e["clap_cnn14"]
print(f"""{tag} (full `tools/final_profile.sh` sequence, one box): **{k(d['value'])} frames/s** ({d['ms_per_step']:.2f} ms/step; \
{k(d['config']['one_batch_at_a_time_frames_per_s'])} one batch at a time), GEMM launches {d['roofline']['achieved']:.0f} TFLOP/s = \
{d['roofline']['frac']:.3f} of peak by HIP events (rocprofv3 of the single-stream pass: QKV {kern('gemm_pp_kernel<0, 128', rk):.1f} µs, \
fc1 {kern('gemm_pp_kernel<1, 160', rk):.1f} µs, residual GEMMs {kern('gemm_pp_kernel<3, 80', rk):.1f} µs, attention \
{kern('attention_kernel<4, false', rk):.1f} µs, LayerNorm {kern('layernorm_kernel<3>', rk):.1f} µs — `{tag}_roofline_kernel_stats.txt`); \
flat search {s['value']:.0f} queries/s at nq=1 (collect kernel {s['roofline']['avg_launch_us'] / 1e3:.2f} ms = {s['roofline']['frac']:.2f} of HBM peak, \
PMC {s['roofline']['traffic'] / 1e6:,.2f} MB per launch), clustered {s['clustered']['queries_per_s']:.0f} \
({s['clustered']['answered_from_the_shadow']}/{s['clustered']['handed_to_fp32_scan']}), nq=256 {k(s['batched_nq256_queries_per_s'])} queries/s \
({s['batched_nq256_roofline']['frac']:.2f} of HBM peak per 128-query pass), nq=32 {k(s['batched_nq32_queries_per_s'])}, nq=4 \
{k(s['batched_nq4_queries_per_s'])}; HTSAT {k(e['clap_htsat']['value'])} clips/s ({e['clap_htsat']['ms_per_step']:.2f} ms/step; \
{e['clap_htsat']['roofline']['frac']:.2f} of HBM peak at kernel boundaries, PMC {e['clap_htsat']['roofline']['traffic'] / 1e9:.2f} GB per forward); \
**MS-CLAP 2022 Cnn14 {k(c['value'])} clips/s at bs=128 × 10 s ({c['ms_per_step']:.2f} ms/step, one batch at a time; \
{k(c['two_batches_in_flight_clips_per_s'])} with two in flight): {c['tflops']:.0f} TFLOP/s over the whole forward = \
{c['frac_of_bf16_peak']:.3f} of the bf16 peak; the convolutions of blocks 2–6 and the head's GEMMs {c['roofline']['achieved']:.0f} TFLOP/s = \
{c['roofline']['frac']:.3f}** (HIP events; per kernel in `{tag}_cnn14_kernel_stats.txt`: fused block 1 {kern('conv_block1', ck) / 1e3:.2f} ms, \
frontend {kern('frontend_kernel', ck) / 1e3:.2f} ms); ViT-L/14 {e['vit_l14']['value'] / 1e3:.2f} k frames/s ({e['vit_l14']['tflops']:.0f} TFLOP/s = \
{e['vit_l14']['frac_of_bf16_peak']:.3f}), ViT-H/14 {e['vit_h14']['value'] / 1e3:.2f} k, ViT-L/16-SigLIP-384 {e['siglip_l16_384']['value'] / 1e3:.2f} k; \
uint8 frames → embeddings {k(e['u8_frames_to_embeddings']['value'])}; CLIP text {e['clip_text_tower']['value'] / 1e3:.0f} k queries/s, one query \
{e['clip_text_tower']['single_query_ms']:.2f} ms; XLM-R-large {k(e['xlmr_text_tower']['value'])}, one query {e['xlmr_text_tower']['single_query_ms']:.2f} ms; \
IVF {k(e['ivf_flat']['value'])}; CPU baselines {d['cpu_baseline']['value']:.0f} frames/s and {s['cpu_baseline']['value']:.2f} queries/s on \
{d['cpu_baseline']['cores']} host threads.""")
