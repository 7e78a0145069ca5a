"""Per-kernel time of one overlay generator forward on a 2550x3300 page (run under rocprofv3 --kernel-trace --stats)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from marie_icr_amd._lib import Context, PREC_F16
from marie_icr_amd.overlay import OverlayModel
from marie_icr_amd.weights import make_overlay_state, make_page_bgr
ctx = Context(0)
m = OverlayModel(ctx, make_overlay_state(0, 64), 64, PREC_F16)
page = torch.from_numpy(make_page_bgr(1, 3300, 2550, n_lines=40)).cuda()
H, W = m.padded_shape(3300, 2550)
fake = torch.empty((H, W, 3), dtype=torch.uint8, device="cuda")
for _ in range(3):
    m.forward_device(page.data_ptr(), 3300, 2550, fake.data_ptr())
torch.cuda.synchronize()
