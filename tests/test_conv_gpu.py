"""conv_igemm through the C ABI vs a plain PyTorch fp32 reference of the same op (F.conv2d + scale/bias +
ReLU + max_pool2d on CPU), over shapes that hit every tail: M not a multiple of 256, N not a multiple of 128,
ragged N (95), odd widths, 2x2 valid filters, 1x1 GEMMs, both pooling modes."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

CASES = [
    # B, H, W, Cin, KH, pad, N, pool, relu, out_f32
    (2, 16, 32, 64, 3, 1, 128, 1, 1, 0),
    (3, 8, 25, 128, 3, 1, 256, 0, 1, 0),     # odd width, M = 600 (tail)
    (2, 8, 25, 64, 3, 1, 64, 2, 1, 0),       # N < 128, 2x1 pool
    (1, 5, 7, 64, 3, 1, 192, 1, 0, 0),       # odd H and W with 2x2 floor pooling
    (2, 2, 26, 128, 2, 0, 128, 0, 1, 0),     # 2x2 valid
    (1, 1, 1, 256, 1, 0, 95, 0, 0, 1),       # single-row GEMM, ragged N, fp32 out
    (700, 1, 1, 512, 1, 0, 2048, 0, 0, 1),   # GEMM, many n-tiles
    (1, 40, 40, 64, 3, 1, 320, 0, 1, 0),     # 3 n-tiles with a partial one
    (1, 1, 25000, 128, 1, 0, 768, 0, 2, 0),  # big plain GEMM + GELU M tail
    (1, 1, 30100, 64, 1, 0, 520, 0, 1, 1),   # big GEMM, fp32 out, partial n-tile (520 = 2 x 256 + 8)
]


def _ref(x, w, scale, bias, pad, pool, relu):
    y = F.conv2d(x.permute(0, 3, 1, 2), w.permute(0, 3, 1, 2), None, padding=pad)
    y = y * scale[None, :, None, None] + bias[None, :, None, None]
    if relu == 2:
        y = F.gelu(y)
    elif relu:
        y = F.relu(y)
    if pool == 1:
        y = F.max_pool2d(y, 2, 2)
    elif pool == 2:
        y = F.max_pool2d(y, (2, 1), (2, 1))
    return y.permute(0, 2, 3, 1).contiguous()


@pytest.fixture(scope="module")
def ctx():
    from marie_icr_amd._lib import Context

    c = Context(0)
    yield c
    c.close()


@pytest.mark.parametrize("prec", ["f32", "f16"])
@pytest.mark.parametrize("case", CASES)
def test_conv_matches_torch(ctx, case, prec):
    from marie_icr_amd._lib import PREC_F16, PREC_F32, ConvDesc

    B, H, W, Cin, K, pad, N, pool, relu, out_f32 = case
    g = torch.Generator().manual_seed(hash(case) & 0xffff)
    x = torch.rand((B, H, W, Cin), generator=g) * 2 - 1
    w = (torch.rand((N, K, K, Cin), generator=g) * 2 - 1) * (3.0 / (K * K * Cin)) ** 0.5
    scale = torch.rand((N,), generator=g) + 0.5
    bias = torch.rand((N,), generator=g) - 0.5
    tdt = torch.float16 if prec == "f16" else torch.float32
    xq, wq = x.to(tdt), w.to(tdt)
    ref = _ref(xq.float(), wq.float(), scale, bias, pad, pool, relu)   # same rounded operands, fp32 math
    dx, dw = xq.cuda(), wq.cuda()
    ds, db = scale.cuda(), bias.cuda()
    odt = torch.float32 if out_f32 else tdt
    out = torch.full(ref.shape, float("nan"), dtype=odt, device="cuda")
    d = ConvDesc(B, H, W, Cin, K, K, pad, N, pool, relu, out_f32, 1, 0)
    ctx.set_stream(torch.cuda.current_stream().cuda_stream)
    ctx.conv2d_nhwc(PREC_F16 if prec == "f16" else PREC_F32, d, dx.data_ptr(), dw.data_ptr(), ds.data_ptr(),
                    db.data_ptr(), out.data_ptr())
    torch.cuda.synchronize()
    got = out.float().cpu()
    assert not torch.isnan(got).any(), "unwritten output elements"
    tol = 2e-5 if (prec == "f32") else (2e-5 if out_f32 else 2e-3)   # f16 output rounding dominates
    err = (got - ref).abs().max().item()
    assert err <= tol * max(1.0, ref.abs().max().item()), (case, prec, err)


def test_conv_rejects_bad_cin(ctx):
    from marie_icr_amd._lib import MarieHipError, PREC_F16, ConvDesc

    x = torch.zeros((1, 4, 4, 48), dtype=torch.float16, device="cuda")
    w = torch.zeros((64, 3, 3, 48), dtype=torch.float16, device="cuda")
    o = torch.zeros((1, 4, 4, 64), dtype=torch.float16, device="cuda")
    with pytest.raises(MarieHipError):
        ctx.conv2d_nhwc(PREC_F16, ConvDesc(1, 4, 4, 48, 3, 3, 1, 64, 0, 0, 0, 1, 0), x.data_ptr(), w.data_ptr(), 0, 0,
                        o.data_ptr())


def test_conv_dilated_and_concat_inputs(ctx):
    """3x3 dilation-6 conv (CRAFT fc6) and the concat-free 1x1 conv over two tensors (CRAFT U-net)."""
    from marie_icr_amd._lib import PREC_F32, ConvDesc

    g = torch.Generator().manual_seed(7)
    ctx.set_stream(torch.cuda.current_stream().cuda_stream)
    # dilated
    x = torch.rand((1, 20, 23, 64), generator=g) * 2 - 1
    w = (torch.rand((96, 3, 3, 64), generator=g) * 2 - 1) * 0.07
    b = torch.rand((96,), generator=g) - 0.5
    ref = F.conv2d(x.permute(0, 3, 1, 2), w.permute(0, 3, 1, 2), b, padding=6, dilation=6).permute(0, 2, 3, 1)
    out = torch.full(ref.shape, float("nan"), device="cuda")
    dx, dw, db = x.cuda(), w.cuda(), b.cuda()        # keep the device tensors alive across the launch
    ctx.conv2d_nhwc(PREC_F32, ConvDesc(1, 20, 23, 64, 3, 3, 6, 96, 0, 0, 0, 6, 0), dx.data_ptr(), dw.data_ptr(), 0,
                    db.data_ptr(), out.data_ptr())
    torch.cuda.synchronize()
    assert (out.cpu() - ref).abs().max().item() <= 2e-5 * max(1.0, ref.abs().max().item())
    # concat of two inputs (64 + 128 channels) -> 1x1
    a = torch.rand((2, 9, 11, 64), generator=g) * 2 - 1
    c = torch.rand((2, 9, 11, 128), generator=g) * 2 - 1
    w = (torch.rand((64, 1, 1, 192), generator=g) * 2 - 1) * 0.1
    cat = torch.cat([a, c], dim=3)
    ref = F.relu(F.conv2d(cat.permute(0, 3, 1, 2), w.permute(0, 3, 1, 2), b[:64])).permute(0, 2, 3, 1)
    out = torch.full(ref.shape, float("nan"), device="cuda")
    da, dc, dw, db = a.cuda(), c.cuda(), w.cuda(), b[:64].cuda()
    ctx.conv2d_nhwc(PREC_F32, ConvDesc(2, 9, 11, 192, 1, 1, 0, 64, 0, 1, 0, 1, 64), da.data_ptr(), dw.data_ptr(), 0,
                    db.data_ptr(), out.data_ptr(), in2_ptr=dc.data_ptr())
    torch.cuda.synchronize()
    assert (out.cpu() - ref).abs().max().item() <= 2e-5 * max(1.0, ref.abs().max().item())



@pytest.mark.parametrize("prec_name,N,ldc,own", [("f16", 45, 48, 1), ("f16", 45, 96, 0), ("f32", 45, 48, 1), ("f32", 13, 40, 0),
                                                   ("f16", 50, 56, 1)])
def test_pitched_ragged_n_gemm(ctx, prec_name, N, ldc, own):
    """A GEMM whose N is not a multiple of 8 written into rows of pitch ldc.  With pad_cols_writable (the logits GEMM: N = 50 265
    in rows of 50 272) the 16-byte store path is taken and the pad columns [N, roundup(N, 8)) come out as zeros; WITHOUT it (a
    GEMM writing a slice of a wider row) the neighbouring columns must be left untouched."""
    import torch

    from marie_icr_amd._lib import PREC_F16, PREC_F32, ConvDesc

    prec = PREC_F16 if prec_name == "f16" else PREC_F32
    dt = torch.float16 if prec_name == "f16" else torch.float32
    M, K = 300, 128
    g = torch.Generator().manual_seed(N * 7 + ldc)
    a = (torch.rand((M, K), generator=g) - 0.5).to(dt)
    w = (torch.rand((N, K), generator=g) - 0.5).to(dt)
    bias = torch.rand((N,), generator=g) - 0.5
    da, dw, db = a.cuda(), w.cuda(), bias.cuda()
    out = torch.full((M, ldc), 7.0, dtype=dt, device="cuda")
    d = ConvDesc(M, 1, 1, K, 1, 1, 0, N, 0, 0, 0, 1, 0, ldc, own)
    ctx.set_stream(torch.cuda.current_stream().cuda_stream)
    ctx.conv2d_nhwc(prec, d, da.data_ptr(), dw.data_ptr(), 0, db.data_ptr(), out.data_ptr())
    torch.cuda.synchronize()
    ref = a.double() @ w.double().T + bias.double()
    got = out.cpu().double()
    tol = 2e-2 if prec_name == "f16" else 1e-4
    assert (got[:, :N] - ref).abs().max() <= tol
    n8 = (N + 7) // 8 * 8
    if own:
        assert (got[:, N:n8] == 0).all()                       # the call's own pad columns: zeros
        assert (got[:, n8:] == 7.0).all()                      # beyond them: untouched
    else:
        assert (got[:, N:] == 7.0).all()                       # a slice of a wider row: nothing outside [0, N) is written
    ctx.set_stream(None)
