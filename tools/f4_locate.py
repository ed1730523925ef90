"""Where does the GPU path leave the oracle on the mini frame + checkpoint (tests/test_data_formats.py, f = 8, hidden 48)?
The same step under the path's switches, each against the oracle run: which tensor moves, under which form.

    python tools/f4_locate.py"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from tests.conftest import load_oracle
from pasco_amd.me import backend
from pasco_amd.graph import fused
import tests.test_data_formats as T

backend.register_checker_backend(load_oracle())
_, exp = T._frame_through_checkpoint("cpu")


def report(tag, got):
    def rel(a, b):
        s = float(b.abs().mean())
        return float((a.cpu() - b).abs().max()) / s
    sem = max(rel(a.F, b.F) for s in exp["sem_logits_at_scales"] for a, b in zip(got["sem_logits_at_scales"][s], exp["sem_logits_at_scales"][s]))
    vox = [rel(a["voxel_logits"].F, b["voxel_logits"].F) for a, b in zip(got["panop_predictions"], exp["panop_predictions"])]
    qry = [rel(a["query_logits"], b["query_logits"]) for a, b in zip(got["panop_predictions"], exp["panop_predictions"])]
    aux = [[rel(x["query_logits"], y["query_logits"]) for x, y in zip(a["aux_outputs"], b["aux_outputs"])]
           for a, b in zip(got["panop_predictions"], exp["panop_predictions"])]
    print(f"{tag:28s} sem {sem:.1e}  voxel {['%.1e' % v for v in vox]}  query {['%.1e' % v for v in qry]}  aux query {[['%.0e' % v for v in a] for a in aux]}")


_, got = T._frame_through_checkpoint("cuda")
report("default (split precision)", got)
fused.set_conv_precision("f32")
_, got = T._frame_through_checkpoint("cuda")
report("exact fp32 MFMA", got)
fused.set_conv_precision("f16x3")
for sw in ("PASCO_HEAD_ABSORB", "PASCO_PE_TABLE", "PASCO_RESIZE_ABSORB", "PASCO_QUERY_GRAPH", "PASCO_MASK_BLOCK", "PASCO_INPUT_FUSED",
           "PASCO_KEEP_FUSED"):
    os.environ[sw] = "0"
    _, got = T._frame_through_checkpoint("cuda")
    report(sw + "=0", got)
    del os.environ[sw]
fused.set_fusion(False)
_, got = T._frame_through_checkpoint("cuda")
report("unfused ME modules", got)
fused.set_fusion(True)
