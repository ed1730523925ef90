"""Wall-clock breakdown of one bench step by phase (synchronising; diagnostic only)."""
import os, sys, time, json
os.environ["PASCO_QUERY_GRAPH"] = "0"     # the synchronising timers cannot sit inside a stream capture
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from pasco_amd.graph.synth import make_scene, TeacherKeep
from pasco_amd.graph import decoder as D, unet as U, transformer as T

dev = torch.device("cuda", 0)
M = int(sys.argv[1]) if len(sys.argv) > 1 else 3
net = bench.build_net(M, 283, dev)
scene = make_scene(0, n_infers=M).to(dev)
tk = TeacherKeep(scene, dev)
times = {}
stack = []
from pasco_amd.graph.profiling import ConvProfiler
from pasco_amd.me.backend import hip_backend
prof = ConvProfiler(); prof.wrap(hip_backend())
marks = []      # (phase name, first record index, one-past-last record index)

def timed(name, fn):
    def w(*a, **k):
        torch.cuda.synchronize(); t = time.perf_counter()
        i0 = len(prof.records)
        r = fn(*a, **k)
        torch.cuda.synchronize(); times[name] = times.get(name, 0.0) + time.perf_counter() - t
        marks.append((name, i0, len(prof.records)))
        return r
    return w

u = net.unet3d
net.prepare_input = timed("prepare_input(pointMLP+merge)", net.prepare_input)
net.ensemble = timed("ensemble(total)", net.ensemble)
net.ensembler.ensemble_sem_compl = timed("  ens_sem", net.ensembler.ensemble_sem_compl)
net.ensembler.ensemble_panop = timed("  ens_panop", net.ensembler.ensemble_panop)
u.encoder.forward = timed("encoder", u.encoder.forward)
u.dense_bottleneck = timed("bottleneck", u.dense_bottleneck)
for i, b in enumerate(u.decoder_generative.dec_blocks):
    b.forward = timed(f"dec_block{i}", b.forward)
u.decoder_generative.predict_panop = timed("predict_panop(total)", u.decoder_generative.predict_panop)
net.transformer_predictor.forward = timed("  transformer", net.transformer_predictor.forward)
tp = net.transformer_predictor
tp.compute_mask_bits = timed("    attn_mask", tp.compute_mask_bits)
for i, l in enumerate(tp.transformer_cross_attention_layers):
    l.forward = timed(f"    xattn{i}", l.forward)
    l.attend = timed(f"    xattn{i}.attend", l.attend)
tp.pred_heads = timed("    pred_heads", tp.pred_heads)
tp.query_step = timed("    query_step(self-attn, ffn, head MLPs)", tp.query_step)
tp.pe_layer.forward = timed("    pos_enc", tp.pe_layer.forward)
for i, l in enumerate(tp.transformer_self_attention_layers):
    l.forward = timed("    self_attn", l.forward)
for i, l in enumerate(tp.transformer_ffn_layers):
    l.forward = timed("    ffn", l.forward)
T.linear_rows = timed("    linear_rows(all)", T.linear_rows)
D.batch_sparse_tensor = timed("  batch_sparse_tensor", D.batch_sparse_tensor)
net.feat._mlp = timed("  point_mlp", net.feat._mlp)
U.merge_subnet_inputs = timed("  merge_subnet_inputs", U.merge_subnet_inputs)
with torch.no_grad():
    for _ in range(2):
        bench.run_scene(net, scene, tk)
    times.clear(); marks.clear(); prof.enabled = True
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(3):
        bench.run_scene(net, scene, tk)
    torch.cuda.synchronize(); tot = (time.perf_counter() - t0) / 3
conv = {}
for name, i0, i1 in marks:
    conv[name] = conv.get(name, 0.0) + sum(r["e0"].elapsed_time(r["e1"]) for r in prof.records[i0:i1])
print("phase: wall ms  (conv + operand-split kernels ms)  other ms")
for k, v in times.items():
    w, c = v / 3 * 1e3, conv.get(k, 0.0) / 3
    print(f"{k:34s} {w:7.2f}  ({c:6.2f})  {w - c:6.2f}")
print("total ms/step (with sync overhead):", round(tot * 1e3, 2))
