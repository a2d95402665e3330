import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from oracle import fixtures as fx
from tests.test_model_gpu import build_model, dev_batch, raw_batch
from tests.util import relmax
from mtn_amd import LabelSmoothing, SimpleLossCompute
dev = torch.device("cuda:0")
name = sys.argv[1] if len(sys.argv) > 1 else "cfg1_query"
dtype = torch.bfloat16 if "bf16" in sys.argv else torch.float32
c = fx.GOLDEN_CONFIGS[name]
g = dict(np.load(f"tests/golden/{name}.npz"))
model = build_model(c, dtype, dev).eval()
b = dev_batch(raw_batch(c), dev)
lc = SimpleLossCompute(model.generator, model.auto_encoder_generator, LabelSmoothing(c["vocab"], fx.PAD, 0.1), opt=None)
ae_y = b.cap if c["auto_encoder_ft"] in ("caption", "summary") else b.query
model.prepare(); model.zero_glue_grads()
out, ae_out = model.forward(b)
loss = lc.loss(out, b.trg_y, b.ntokens, ae_out, ae_y, (ae_y != fx.PAD).sum())
loss.backward()
torch.cuda.synchronize()
print("loss", float(loss), float(g["loss"]))
norms = dict(zip([str(s) for s in g["grad_names"]], g["grad_norms"]))
params = dict(model.named_parameters())
bad = []
for k, n in norms.items():
    gr = params[k].grad
    if "grad." + k in g: e = relmax(gr, torch.from_numpy(g["grad." + k]))
    else: e = relmax(gr.reshape(-1)[:256], torch.from_numpy(g["gradhead." + k]))
    gn = float(gr.double().norm())
    if e > 1e-3 or abs(gn - n) / max(n, 1e-9) > 1e-3:
        bad.append((k, e, gn, n))
print(len(bad), "bad of", len(norms))
for x in bad[:60]: print(x)
