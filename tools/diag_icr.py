import os, sys, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from marie_icr_amd._lib import Context, PREC_F32, PREC_F16
from marie_icr_amd.icr import IcrModel
from marie_icr_amd.weights import make_icr_state
ctx = Context(0)
for tag in "ab":
    g = np.load(f"tests/golden/icr_attn_{tag}.npz")
    st = make_icr_state(int(g["weight_seed"]))
    for prec in (PREC_F32, PREC_F16):
        m = IcrModel(ctx, st, precision=prec)
        out = m.forward_host(g["crops_u8"], want_logits=True, want_rectified=True)
        print(tag, prec, "rect", np.abs(out["rectified"] - g["rectified"][:, 0]).max(),
              "logit0", np.abs(out["logits"][:, 0] - g["logits"][:, 0]).max(),
              "all", np.abs(out["logits"] - g["logits"]).max(axis=(1, 2)), "range", np.abs(g["logits"]).max(),
              "argmax eq", (out["argmax"] == g["argmax"]).all(axis=1))
        m.close()
