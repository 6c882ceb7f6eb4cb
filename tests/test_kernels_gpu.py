"""Op-level parity: every hand-written gfx950 kernel family (through the C ABI, include/distdiff_hip_ops.h)
against a plain torch fp32 CPU computation of the same operation on the same bf16-rounded inputs.

Tolerance statement: inputs/weights are rounded to bf16 on both sides, the kernels accumulate in fp32 and
round the result to bf16 once, so |err| <= ~2^-8 * |ref| + accumulation noise; checked as
max|err| <= atol + rtol*max|ref| with the per-test values below.
"""
import ctypes as C
import os
import math

import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu


def bf(x):
    return x.to(torch.bfloat16).float()


def assert_close(got, ref, rtol=1.5e-2, atol=1e-3, what=""):
    got = got.float().cpu()
    ref = ref.float().cpu()
    assert got.shape == ref.shape, (what, got.shape, ref.shape)
    assert torch.isfinite(got).all(), what + ": non-finite output"
    err = (got - ref).abs().max().item()
    lim = atol + rtol * ref.abs().max().item()
    assert err <= lim, "%s: max err %.4g > %.4g (ref max %.4g)" % (what, err, lim, ref.abs().max().item())


@pytest.fixture(scope="module")
def ops(hip_lib):
    from distdiff_amd import ops as o
    assert torch.cuda.is_available()
    return o


CONV_CASES = [
    # name, B, Cin, Cout, H, W, k, stride, pad, up
    ("3x3_c64", 2, 64, 128, 16, 16, 3, 1, 1, 0),
    ("3x3_c320_narrowN", 2, 320, 320, 16, 16, 3, 1, 1, 0),
    ("3x3_smallcin", 2, 4, 64, 16, 16, 3, 1, 1, 0),
    ("3x3_cin40_ragged", 1, 40, 72, 9, 7, 3, 1, 1, 0),
    ("3x3_stride2", 2, 64, 64, 16, 16, 3, 2, 1, 0),
    ("3x3_up2", 2, 64, 64, 8, 8, 3, 1, 1, 1),
    ("1x1", 2, 128, 256, 8, 8, 1, 1, 0, 0),
    ("1x1_stride2", 2, 64, 128, 8, 8, 1, 2, 0, 0),
    ("7x7_stem", 1, 3, 64, 32, 32, 7, 2, 3, 0),
    ("3x3_cout3", 1, 128, 3, 16, 16, 3, 1, 1, 0),
    ("3x3_deepK_splitk", 2, 1280, 128, 8, 8, 3, 1, 1, 0),
    # shapes routed to the persistent big-tile kernel (M >= 1024): 128x256, 128x320, 256x128 tiles, ragged M, split-K
    ("big_128x256", 2, 128, 256, 32, 32, 3, 1, 1, 0),
    ("big_128x320_ragged", 2, 320, 320, 24, 25, 3, 1, 1, 0),
    ("big_256x128", 1, 64, 128, 40, 40, 3, 1, 1, 0),
    ("big_up2", 2, 64, 256, 16, 16, 3, 1, 1, 1),
    ("big_stride2", 2, 64, 256, 64, 64, 3, 2, 1, 0),
    ("big_1x1_shallowK", 2, 320, 640, 32, 32, 1, 1, 0, 0),
    ("big_deepK_splitk", 2, 1280, 320, 24, 24, 3, 1, 1, 0),
    # <= 100 (N % 160 == 0) / <= 24 (N % 128 == 0) K-steps run as two 4-wave workgroups per CU (128x160 / 128x128, 2 LDS stages);
    # deeper items keep the 8-wave 128x256 / 256x160 / 256x128 forms: one case per form and per epilogue variant
    ("big8_128x256_deepK", 2, 512, 256, 24, 24, 3, 1, 1, 0),
    ("big8_256x128_deepK", 2, 512, 128, 32, 32, 3, 1, 1, 0),
    ("big8_256x160_100steps", 4, 768, 160, 32, 32, 3, 1, 1, 0),
    ("big4_128x160_pointwise", 4, 1280, 320, 32, 32, 1, 1, 0, 0),
    ("big4_128x128_ragged", 3, 128, 384, 20, 21, 3, 1, 1, 0),
    # 3x3 / stride 1 shapes with >= 192 tiles of 256 pixels, W a power of two, Cin % 64 == 0, N % 320 == 0 or % 256 == 0: the halo-resident
    # ping-pong kernel (conv_halo.hip), forward and -- where Cin qualifies as N -- input-gradient; image widths 16 / 32 / 64 / 128
    ("halo_320_32x32", 48, 320, 320, 32, 32, 3, 1, 1, 0),
    ("halo_256_64x64", 12, 64, 256, 64, 64, 3, 1, 1, 0),
    ("halo_640_16x16", 192, 128, 640, 16, 16, 3, 1, 1, 0),
    ("halo_256_128x128", 3, 256, 256, 128, 128, 3, 1, 1, 0),
    ("halo_up2_320", 24, 128, 320, 32, 32, 3, 1, 1, 1),          # fused nearest-2x upsample: logical 64x64
    ("halo_w256_2x128_tiles", 1, 64, 256, 256, 256, 3, 1, 1, 0),  # images wider than 128: 2 x 128-pixel tiles
    ("halo_w512_up2", 1, 64, 256, 64, 256, 3, 1, 1, 1),           # logical 128 x 512
    ("halo_n128_512x128_tiles", 1, 128, 128, 256, 512, 3, 1, 1, 0),  # N = 128: 512-row tiles (4 x 128 pixels), waves 4 x 2
    # the PERSISTENT 512 x 128 form (conv_halo_persist_kernel: decoder levels, <= 8 chunks): workgroups that walk 2 - 3 tiles with the next
    # tile's halo / first weight stage requested under the epilogue; an odd number of K-steps per tile (1 and 3 chunks: the weight stage
    # parity flips from tile to tile), two n-tiles per pixel tile, a ragged tile count (288 tiles on 256 workgroups), fused upsample
    ("halo_persist_n128_cin64_3tiles_oddK", 6, 64, 128, 256, 256, 3, 1, 1, 0),
    ("halo_persist_n256_cin192_2ntiles", 3, 192, 256, 256, 256, 3, 1, 1, 0),
    ("halo_persist_ragged_288_tiles", 9, 128, 128, 128, 128, 3, 1, 1, 0),
    ("halo_persist_up2_cin128", 4, 128, 128, 128, 128, 3, 1, 1, 1),
    # the 8 x 8 level (M = 64 pixels x images): the halo-resident kernel with four whole images per 256-pixel tile and a split over the
    # 64-channel chunks into fp32 partial sums + the reduce kernel (4-way at 64 images, 8-way at 32), forward and input-gradient
    ("halo_8x8_multi_image_split4", 64, 1280, 1280, 8, 8, 3, 1, 1, 0),
    ("halo_8x8_cin2560_split8", 32, 2560, 1280, 8, 8, 3, 1, 1, 0),
    # 16 images (the automatic batch of a small-HBM device): 16 tiles want a 16-way split, 40 chunks do not divide by 16 -> 8-way
    ("halo_8x8_cin2560_b16_split_divides_chunks", 16, 2560, 1280, 8, 8, 3, 1, 1, 0),
    # >= 192 tiles at 8 x 8 (--resolution 256 with a large engine batch, or an ABI caller): no chunk split, so no multi-image halo form --
    # the general kernels take it (was: hipErrorInvalidValue from the halo launcher)
    ("8x8_many_tiles_no_split", 192, 128, 1280, 8, 8, 3, 1, 1, 0),
    # N <= 4 (conv_out of the decoder 128 -> 3 and of the UNet 320 -> 4): the small kernel's 256 x 64 tiles
    ("n3_vae_conv_out", 1, 128, 3, 256, 256, 3, 1, 1, 0),
    ("n4_unet_conv_out", 16, 320, 4, 64, 64, 3, 1, 1, 0),
]


@pytest.mark.parametrize("case", CONV_CASES, ids=[c[0] for c in CONV_CASES])
def test_conv_forward_and_dgrad(ops, case):
    name, B, Cin, Cout, H, W, k, stride, pad, up = case
    g = torch.Generator().manual_seed(hash(name) % 1000)
    x = bf(torch.randn(B, Cin, H, W, generator=g))
    w = bf(torch.randn(Cout, Cin, k, k, generator=g) / math.sqrt(Cin * k * k))
    bias = torch.randn(Cout, generator=g)
    xin = F.interpolate(x, scale_factor=2, mode="nearest") if up else x
    x_req = xin.clone().requires_grad_(True)
    ref = F.conv2d(x_req, w, bias, stride=stride, padding=pad)
    Ho, Wo = ref.shape[2], ref.shape[3]
    cin_pad = (Cin + 7) // 8 * 8
    xd = ops.to_nhwc_bf16(x, cin_pad).cuda()
    pk = ops.PackedConv(w, pad, mode=0, bias=bias)
    y = ops.conv_gemm(xd, pk, B, H, W, Ho, Wo, stride=stride, shift=up)
    torch.cuda.synchronize()
    assert_close(ops.from_nhwc(y, B, Ho, Wo), ref.detach(), what=name + " fwd")
    # input-gradient (dgrad): conv of dY with the transposed/flipped weights; stride 2 -> input-dilated gather
    dy = bf(torch.randn(B, Cout, Ho, Wo, generator=g))
    (gref,) = torch.autograd.grad(ref, x_req, dy)
    cout_pad = (Cout + 7) // 8 * 8
    dyd = ops.to_nhwc_bf16(dy, cout_pad).cuda()
    pkd = ops.PackedConv(w, pad, mode=1)
    Hl, Wl = xin.shape[2], xin.shape[3]
    if stride == 2:
        dx = ops.conv_gemm(dyd, pkd, B, Ho, Wo, Hl, Wl, stride=1, shift=1, parity=1)
    else:
        dx = ops.conv_gemm(dyd, pkd, B, Ho, Wo, Hl, Wl, stride=1)
    torch.cuda.synchronize()
    assert_close(ops.from_nhwc(dx, B, Hl, Wl), gref, what=name + " dgrad")
    if up:  # transpose of the fused nearest-2x: 2x2 sum pooling
        from distdiff_amd import _lib
        dxs = torch.empty((B * H * W, dx.shape[1]), device="cuda", dtype=torch.bfloat16)
        _lib.check(_lib.lib().dd_op_sumpool2x2(C.c_void_p(dx.data_ptr()), dx.stride(0), C.c_void_p(dxs.data_ptr()), dxs.stride(0),
                                               B, H, W, dx.shape[1], 0, None))
        torch.cuda.synchronize()
        xr = x.clone().requires_grad_(True)
        r2 = F.conv2d(F.interpolate(xr, scale_factor=2, mode="nearest"), w, bias, stride=stride, padding=pad)
        (g2,) = torch.autograd.grad(r2, xr, dy)
        assert_close(ops.from_nhwc(dxs, B, H, W), g2, rtol=2.5e-2, what=name + " dgrad+sumpool")


def test_narrow_conv_fp32_output_in_padded_rows(ops):
    """The decoder's conv_out as the engine calls it: fp32 output into 8-wide rows (only the N valid channels are written), input rows
    wider than cin (a column view), image borders on all four sides of a non-square image."""
    g = torch.Generator().manual_seed(31)
    B, Cin, Cout, H, W = 2, 128, 3, 96, 352
    x = bf(torch.randn(B, Cin, H, W, generator=g))
    w = bf(torch.randn(Cout, Cin, 3, 3, generator=g) / math.sqrt(Cin * 9))
    bias = torch.randn(Cout, generator=g)
    ref = F.conv2d(x, w, bias, padding=1)
    xs = torch.zeros((B * H * W, Cin + 64), dtype=torch.bfloat16, device="cuda")
    xs[:, 64:] = ops.to_nhwc_bf16(x, Cin).cuda()
    y = torch.full((B * H * W, 8), 7.0, dtype=torch.float32, device="cuda")
    pk = ops.PackedConv(w, 1, mode=0, bias=bias)
    ops.conv_gemm(xs[:, 64:], pk, B, H, W, H, W, y=y[:, :Cout], out_f32=True, ksplit=1, x_ld=xs.stride(0))
    torch.cuda.synchronize()
    assert_close(ops.from_nhwc(y[:, :Cout], B, H, W), ref, rtol=5e-3, atol=1e-3, what="narrow conv fp32")
    assert float((y[:, Cout:] - 7.0).abs().max()) == 0.0, "wrote outside the N valid channels"


@pytest.mark.parametrize("case", [("decoder_conv_out_128to3_512px", 1, 128, 3, 512, 512, True), ("unet_conv_out_320to4_64px", 32, 320, 4, 64, 64, True),
                                  ("narrow_bf16_out_64to2_128px", 8, 64, 2, 128, 128, False)], ids=lambda c: c[0])
def test_narrow_conv_on_the_halo_form(ops, case):
    """conv_out (N <= 4) on the 512 x 32 form of the halo kernel (conv_halo_kernel<1, 2>): the input tile is staged once per 64-channel
    chunk, the weight stage is padded with out-of-range (zero) rows; fp32 output into 8-wide rows touches only the N valid channels."""
    name, B, Cin, Cout, H, W, f32 = case
    g = torch.Generator().manual_seed(33)
    x = bf(torch.randn(B, Cin, H, W, generator=g))
    w = bf(torch.randn(Cout, Cin, 3, 3, generator=g) / math.sqrt(Cin * 9))
    bias = torch.randn(Cout, generator=g)
    ref = F.conv2d(x, w, bias, padding=1)
    xs = ops.to_nhwc_bf16(x, Cin).cuda()
    pk = ops.PackedConv(w, 1, mode=0, bias=bias)
    if f32:
        y = torch.full((B * H * W, 8), 7.0, dtype=torch.float32, device="cuda")
        ops.conv_gemm(xs, pk, B, H, W, H, W, y=y[:, :Cout], out_f32=True, ksplit=1)
    else:
        y = torch.full((B * H * W, 8), 7.0, dtype=torch.bfloat16, device="cuda")
        ops.conv_gemm(xs, pk, B, H, W, H, W, y=y[:, :Cout], ksplit=1)
    torch.cuda.synchronize()
    assert_close(ops.from_nhwc(y[:, :Cout].float(), B, H, W), ref, rtol=5e-3 if f32 else 2e-2, atol=1e-3 if f32 else 2e-2, what=name)
    assert float((y[:, Cout:].float() - 7.0).abs().max()) == 0.0, "wrote outside the N valid channels"


def test_conv_general_staging_path_beyond_4gb(ops):
    """The decoder's 256-channel 512x512 level at the benchmarked batch: 34 x 512 x 512 x 256 bf16 = 4.56 GB of input.  The fast staging
    path of conv_gemm2.hip addresses the input through 32-bit BYTE offsets of a buffer resource, so inputs of 3.75 GB and more take the
    general path (32-bit element offsets, global_load_lds): images before, across and beyond the 2^32-byte boundary are checked
    against F.conv2d on the CPU, forward (256 -> 128, 3x3) and through the residual / bias epilogue."""
    free, _ = torch.cuda.mem_get_info()
    if free < 12e9:
        pytest.skip("needs 12 GB of free HBM")
    B, Cin, Cout, H, W = 34, 256, 128, 512, 512
    g = torch.Generator().manual_seed(99)
    w = bf(torch.randn(Cout, Cin, 3, 3, generator=g) / math.sqrt(Cin * 9))
    bias = torch.randn(Cout, generator=g)
    pk = ops.PackedConv(w, 1, mode=0, bias=bias)
    gg = torch.Generator(device="cuda").manual_seed(5)
    x = torch.empty((B * H * W, Cin), device="cuda", dtype=torch.bfloat16)
    for b in range(B):       # image by image: torch.randn of 2.2e9 elements at once would need a 9 GB fp32 temporary
        x[b * H * W:(b + 1) * H * W] = torch.randn((H * W, Cin), device="cuda", generator=gg).to(torch.bfloat16)
    assert x.numel() * 2 >= 0xF0000000
    res = torch.empty((B * H * W, Cout), device="cuda", dtype=torch.bfloat16)
    res.copy_(x[:, :Cout])
    y = ops.conv_gemm(x, pk, B, H, W, H, W, res=res, ksplit=1)
    torch.cuda.synchronize()
    for b in (0, 31, 32, 33):     # image 32 starts exactly at byte 2^32 of the input
        xi = x[b * H * W:(b + 1) * H * W].float().cpu().reshape(1, H, W, Cin).permute(0, 3, 1, 2)
        ref = F.conv2d(xi, w, bias, padding=1) + xi[:, :Cout]
        got = y[b * H * W:(b + 1) * H * W].float().cpu().reshape(1, H, W, Cout).permute(0, 3, 1, 2)
        assert_close(got, ref, what="general path, image %d" % b)


@pytest.mark.parametrize("case", [("gnfold_320_32x32", 48, 320, 320, 32, 32, 32, True), ("gnfold_256_64x64_res", 12, 128, 256, 64, 64, 32, True),
                                  ("gnfold_n128_nosilu", 1, 64, 128, 256, 512, 32, False),
                                  ("gnfold_conv_out_128to3_512px", 1, 128, 3, 512, 512, 32, True), ("gnfold_conv_out_320to4_64px", 32, 320, 4, 64, 64, 32, True)],
                         ids=lambda c: c[0])
def test_groupnorm_applied_by_the_halo_convolution(ops, case):
    """CF_GNFOLD: GroupNorm(+SiLU) -> 3x3 convolution (ResnetBlock2D norm1 -> conv1, norm2 -> conv2) with the normalisation applied to
    the convolution's staged input tile instead of a pass over the tensor: same result as GroupNorm -> conv through the separate
    kernels (the affine, SiLU and bf16 rounding are the same arithmetic) and as torch on the same bf16 inputs."""
    name, B, Cin, Cout, H, W, G, silu = case
    g = torch.Generator().manual_seed(len(name))
    x = bf(torch.randn(B, Cin, H, W, generator=g) * 1.5 + 0.4)
    gamma, beta = 1.0 + 0.2 * torch.randn(Cin, generator=g), 0.3 * torch.randn(Cin, generator=g)
    w = bf(torch.randn(Cout, Cin, 3, 3, generator=g) / math.sqrt(Cin * 9))
    bias = torch.randn(Cout, generator=g)
    hn = F.group_norm(x, G, gamma, beta, 1e-5)
    ref = F.conv2d(bf(F.silu(hn) if silu else hn), w, bias, padding=1)
    xd = ops.to_nhwc_bf16(x, Cin).cuda()
    pk = ops.PackedConv(w, 1, mode=0, bias=bias)
    coef, stats = ops.groupnorm_coef(xd, gamma.cuda(), beta.cuda(), B, H * W, G, 1e-5)
    yn, _ = ops.groupnorm(xd, gamma.cuda(), beta.cuda(), B, H * W, G, 1e-5, silu)
    if Cout <= 4:      # conv_norm_out -> conv_out as the engine runs it: fp32 output into 8-wide rows, the narrow form of the halo kernel
        y = torch.zeros((B * H * W, 8), dtype=torch.float32, device="cuda")
        ops.conv_gemm(xd, pk, B, H, W, H, W, y=y[:, :Cout], out_f32=True, gn_coef=coef, gn_silu=silu, ksplit=1)
        y2 = torch.zeros((B * H * W, 8), dtype=torch.float32, device="cuda")
        ops.conv_gemm(yn, pk, B, H, W, H, W, y=y2[:, :Cout], out_f32=True, ksplit=1)
        torch.cuda.synchronize()
        assert_close(ops.from_nhwc(y[:, :Cout], B, H, W), ref, rtol=5e-3, atol=2e-3, what=name + " vs torch")
        assert torch.equal(y.cpu(), y2.cpu()), "folded and separate GroupNorm -> conv differ bitwise"
        return
    y = ops.conv_gemm(xd, pk, B, H, W, H, W, gn_coef=coef, gn_silu=silu)
    torch.cuda.synchronize()
    assert_close(ops.from_nhwc(y, B, H, W), ref, what=name + " vs torch")
    y2 = ops.conv_gemm(yn, pk, B, H, W, H, W)
    torch.cuda.synchronize()
    assert torch.equal(y.cpu(), y2.cpu()), "folded and separate GroupNorm -> conv differ bitwise"


def test_linear_epilogues(ops):
    g = torch.Generator().manual_seed(3)
    M, K, N = 300, 320, 640
    x = bf(torch.randn(M, K, generator=g))
    w = bf(torch.randn(N, K, generator=g) / math.sqrt(K))
    bias = torch.randn(N, generator=g)
    res = bf(torch.randn(M, N, generator=g))
    xd = x.to(torch.bfloat16).cuda()
    pk = ops.PackedConv(w, 0, bias=bias)
    ref = x @ w.t() + bias
    for ks in (1, 0, 3):
        y = ops.conv_gemm(xd, pk, 1, M, 1, M, 1, ksplit=ks)
        assert_close(y, ref, what="linear ksplit=%d" % ks)
    y = ops.conv_gemm(xd, pk, 1, M, 1, M, 1, res=res.to(torch.bfloat16).cuda(), ksplit=1)
    assert_close(y, ref + res, what="linear+res")
    y = ops.conv_gemm(xd, pk, 1, M, 1, M, 1, relu=True, out_f32=True, ksplit=2)
    assert y.dtype == torch.float32
    assert_close(y, ref.clamp_min(0), rtol=5e-3, what="linear relu f32 splitk")
    msk = torch.randn(M, N, generator=g)
    y = ops.conv_gemm(xd, pk, 1, M, 1, M, 1, mask=msk.to(torch.bfloat16).cuda(), ksplit=1)
    assert_close(y, ref * (bf(msk) > 0), what="linear mask")
    # strided output / input views (channel concat as a view)
    big = torch.zeros((M, N + 64), device="cuda", dtype=torch.bfloat16)
    ops.conv_gemm(xd, pk, 1, M, 1, M, 1, y=big[:, 64:], ksplit=1)
    assert_close(big[:, 64:], ref, what="linear strided out")
    assert float(big[:, :64].float().abs().max()) == 0.0


def test_pointwise_launch_requires_the_centre_tap(ops):
    """dd_op_conv_gemm: a one-tap, stride-1, same-size launch is a 1x1 / linear layer; the persistent kernel does not read its tap
    table.  The asynchronous entry point never reads device memory: ops.conv_gemm checks the READ-ONLY host copy kept at pack time, and
    dd_op_conv_gemm_check is the synchronous companion that validates the DEVICE table (a table that diverged from the host copy)."""
    g = torch.Generator().manual_seed(12)
    M, K, N = 2048, 256, 256
    x = bf(torch.randn(M, K, generator=g)).to(torch.bfloat16).cuda()
    pk = ops.PackedConv(bf(torch.randn(N, K, generator=g) / 16), 0)
    assert int(pk.taptab[0]) == (32 << 6) | 32 == int(pk.taptab_host[0])
    ops.conv_gemm(x, pk, 1, M, 1, M, 1, ksplit=1, check_device_taps=True)
    with pytest.raises(ValueError):
        pk.taptab_host[0] = 0                                          # the host copy cannot be edited
    pk.taptab[0] = ((32 + 1) << 6) | 32                                # the device table diverges: dy = +1
    ops.conv_gemm(x, pk, 1, M, 1, M, 1, ksplit=1)                      # the asynchronous launch cannot see it (documented contract) ...
    with pytest.raises(RuntimeError):
        ops.conv_gemm(x, pk, 1, M, 1, M, 1, ksplit=1, check_device_taps=True)     # ... the synchronous check does
    # a host copy with a non-centre single tap is refused without touching the device
    pk2 = ops.PackedConv(bf(torch.randn(N, K, generator=g) / 16), 0)
    bad = pk2.taptab_host.copy()
    bad[0] = ((32 + 1) << 6) | 32
    bad.setflags(write=False)
    pk2.taptab_host = bad
    with pytest.raises(RuntimeError):
        ops.conv_gemm(x, pk2, 1, M, 1, M, 1, ksplit=1)
    # 3x3 tables pass the device check
    pk3 = ops.PackedConv(bf(torch.randn(64, 64, 3, 3, generator=g) / 24), 1)
    x3 = bf(torch.randn(1024, 64, generator=g)).to(torch.bfloat16).cuda()
    ops.conv_gemm(x3, pk3, 1, 32, 32, 32, 32, check_device_taps=True)


@pytest.mark.parametrize("M,K,Fd", [(200, 128, 256), (1500, 320, 1280), (12288, 320, 512), (16384, 640, 1280)])   # the last two: persistent ping-pong GEMM, 256 x 256 tiles
def test_geglu_forward_backward(ops, M, K, Fd):
    from distdiff_amd import _lib
    g = torch.Generator().manual_seed(4)
    x = bf(torch.randn(M, K, generator=g))
    w = bf(torch.randn(2 * Fd, K, generator=g) / math.sqrt(K))
    bias = torch.randn(2 * Fd, generator=g) * 0.1
    pk = ops.PackedConv(w, 0, geglu=True, bias=bias)
    xd = x.to(torch.bfloat16).cuda()
    raw = torch.zeros((M, 2 * Fd), device="cuda", dtype=torch.bfloat16)
    xr = x.clone().requires_grad_(True)
    proj = xr @ w.t() + bias
    hid, gate = proj.chunk(2, dim=-1)
    ref = hid * F.gelu(gate)
    for ks in (1, 2):
        y = ops.conv_gemm(xd, pk, 1, M, 1, M, 1, raw=raw, ksplit=ks)
        assert_close(y, ref.detach(), rtol=2e-2, what="geglu fwd ks=%d" % ks)
    dout = bf(torch.randn(M, Fd, generator=g))
    (gx,) = torch.autograd.grad(ref, xr, dout)
    draw = torch.zeros_like(raw)
    dd = dout.to(torch.bfloat16).cuda()
    _lib.check(_lib.lib().dd_op_geglu_bwd(C.c_void_p(raw.data_ptr()), raw.stride(0), C.c_void_p(dd.data_ptr()), dd.stride(0),
                                          C.c_void_p(draw.data_ptr()), draw.stride(0), M, Fd, None))
    pkd = ops.PackedConv(w, 0, mode=1, geglu=True)
    dx = ops.conv_gemm(draw, pkd, 1, M, 1, M, 1, ksplit=1)
    assert_close(dx, gx, rtol=3e-2, what="geglu dgrad")


@pytest.mark.parametrize("Cc,G,HW,silu,eps", [(320, 32, 256, True, 1e-5), (128, 32, 1024, True, 1e-6), (64, 8, 100, False, 1e-6),
                                               (2560, 32, 64, True, 1e-5)])
def test_groupnorm(ops, Cc, G, HW, silu, eps):
    g = torch.Generator().manual_seed(5)
    B = 2
    x = bf(torch.randn(B, Cc, HW, generator=g) * 2 + 0.5)
    gamma = torch.randn(Cc, generator=g)
    beta = torch.randn(Cc, generator=g)
    xr = x.clone().requires_grad_(True)
    ref = F.group_norm(xr, G, gamma, beta, eps)
    if silu:
        ref = F.silu(ref)
    xd = x.permute(0, 2, 1).reshape(B * HW, Cc).to(torch.bfloat16).cuda()
    y, stats = ops.groupnorm(xd, gamma.cuda(), beta.cuda(), B, HW, G, eps, silu)
    assert_close(y.float().cpu().reshape(B, HW, Cc).permute(0, 2, 1), ref.detach(), what="gn fwd")
    dy = bf(torch.randn(B, Cc, HW, generator=g))
    (gx,) = torch.autograd.grad(ref, xr, dy)
    dyd = dy.permute(0, 2, 1).reshape(B * HW, Cc).to(torch.bfloat16).cuda()
    dx = ops.groupnorm(xd, gamma.cuda(), beta.cuda(), B, HW, G, eps, silu, dy=dyd, stats=stats)
    assert_close(dx.float().cpu().reshape(B, HW, Cc).permute(0, 2, 1), gx, rtol=2e-2, atol=2e-3, what="gn bwd")


@pytest.mark.parametrize("Cc", [320, 640, 1280, 64, 2048])     # 1-3 vectors per lane (multi-row kernel), one-row kernel beyond 1536
def test_layernorm(ops, Cc):
    g = torch.Generator().manual_seed(6)
    M = 77
    x = bf(torch.randn(M, Cc, generator=g) * 1.5 + 0.3)
    gamma, beta = torch.randn(Cc, generator=g), torch.randn(Cc, generator=g)
    xr = x.clone().requires_grad_(True)
    ref = F.layer_norm(xr, (Cc,), gamma, beta, 1e-5)
    xd = x.to(torch.bfloat16).cuda()
    y, stats = ops.layernorm(xd, gamma.cuda(), beta.cuda(), 1e-5)
    assert_close(y, ref.detach(), what="ln fwd")
    dy = bf(torch.randn(M, Cc, generator=g))
    (gx,) = torch.autograd.grad(ref, xr, dy)
    dx = ops.layernorm(xd, gamma.cuda(), beta.cuda(), 1e-5, dy=dy.to(torch.bfloat16).cuda(), stats=stats)
    assert_close(dx, gx, rtol=2e-2, atol=2e-3, what="ln bwd")


@pytest.mark.parametrize("M,Cc,N,geglu", [(4096, 320, 960, False), (2048, 640, 640, False), (4096, 320, 2560, True), (300, 320, 320, False),
                                            (1024, 1280, 3840, False), (49152, 320, 960, False), (49152, 640, 640, False),
                                            (12288, 320, 1024, True), (24576, 640, 2560, True)])
def test_layernorm_folded_into_linear(ops, M, Cc, N, geglu):
    """CF_LNFOLD: LayerNorm(x) W^T + b computed as rstd * (x (gamma o W)^T - mean * c1) + (b + W beta) from the RAW x and the row
    statistics -- the transformer blocks' LayerNorm -> QKV / to_q / GEGLU projections (diffusers BasicTransformerBlock).
    Checked against torch's layer_norm + linear on the same bf16 inputs (the folded form never rounds LayerNorm(x) to bf16, so it is
    the more accurate of the two); M = 300 takes the small kernel's generic epilogue, M = 49152 (>= 192 tiles of 256 rows, N <= 1280) the
    ping-pong GEMM of conv_halo.hip (GEGLU: its 256 x 256 form, a ragged number of tiles per persistent workgroup), the others the batched
    two-workgroup forms."""
    g = torch.Generator().manual_seed(77)
    x = bf(torch.randn(M, Cc, generator=g) * 1.7 + torch.randn(M, 1, generator=g) * 0.8)      # row means of the size of the row std
    gamma, beta = 1.0 + 0.3 * torch.randn(Cc, generator=g), 0.2 * torch.randn(Cc, generator=g)
    w = torch.randn(N, Cc, generator=g) / math.sqrt(Cc)
    bias = torch.randn(N, generator=g)
    ref = F.linear(F.layer_norm(x, (Cc,), gamma, beta, 1e-5), bf(w), bias)
    if geglu:
        hh, gg = ref.chunk(2, dim=-1)
        ref = hh * F.gelu(gg)
    wf = bf(w * gamma[None, :])
    bfold = bias + bf(w) @ beta
    pk = ops.PackedConv(wf, 0, geglu=geglu, bias=bfold)
    pc = ops.PackedConv(wf, 0, geglu=geglu, bias=wf.sum(dim=1))          # c1 in the packed column order (reuses the bias permutation)
    xd = x.to(torch.bfloat16).cuda()
    stats = ops.layernorm_stats(xd, 1e-5)
    mu, var = x.mean(1), x.var(1, unbiased=False)
    assert_close(stats[:, 0], mu, rtol=1e-4, atol=1e-5, what="ln mean")
    assert_close(stats[:, 1], (var + 1e-5).rsqrt(), rtol=1e-4, atol=1e-5, what="ln rstd")
    y = ops.conv_gemm(xd, pk, 1, M, 1, M, 1, ksplit=1, ln_stats=stats, ln_c1=pc.bias)
    torch.cuda.synchronize()
    assert_close(y, ref, rtol=1.5e-2, atol=2e-2, what="ln-folded linear")


WS_CASES = [("n320_bias", 36928, 320, False, False, False, False), ("n320_rowstats", 36928, 320, False, False, True, False), ("n320_res_rowstats_takes_the_pingpong_gemm", 36864, 320, True, False, True, False),
            ("n960_lnfold", 33088, 960, False, True, False, False), ("n640_rowstats_views", 32768, 640, False, False, True, False),
            ("geglu2560_lnfold_raw", 33088, 2560, False, True, False, True), ("geglu512_plain", 32832, 512, False, False, False, True)]


@pytest.mark.parametrize("case", WS_CASES, ids=[c[0] for c in WS_CASES])
def test_weight_stationary_gemm(ops, case):
    """gemm_ws.hip: the K = 320 pointwise layers of the transformer blocks at the 64x64 level (fused QKV, to_q, to_out + residual + row
    statistics, proj_in, the GEGLU projection; diffusers BasicTransformerBlock) with the weights held in registers,
    K split over the two waves of a SIMD and 64-row activation tiles streamed through LDS.  Ragged numbers of row tiles per workgroup,
    1 / 2 / 3 / 10 column blocks, strided input / output / residual views, every epilogue flag; against torch on the same bf16 inputs."""
    _, M, N, res, lnfold, rowstats, geglu = case
    K = 320
    g = torch.Generator().manual_seed(79)
    x = bf(torch.randn(M, K, generator=g) * 1.3 + (torch.randn(M, 1, generator=g) * 0.6 if lnfold else 0.0))
    w = torch.randn(N, K, generator=g) / math.sqrt(K)
    bias = torch.randn(N, generator=g)
    views = "views" in case[0]
    xs = torch.zeros((M, K + (64 if views else 0)), dtype=torch.bfloat16, device="cuda")
    xs[:, -K:] = x.to(torch.bfloat16).cuda()
    xd = xs[:, -K:]
    ncols = N // 2 if geglu else N
    ys = torch.zeros((M, ncols + (32 if views else 0)), dtype=torch.bfloat16, device="cuda")
    yd = ys[:, :ncols]
    rd = None
    if res:
        r = bf(torch.randn(M, N, generator=g) * 1.5 + 0.3)
        rs_ = torch.zeros((M, N + (8 if views else 0)), dtype=torch.bfloat16, device="cuda")
        rs_[:, :N] = r.to(torch.bfloat16).cuda()
        rd = rs_[:, :N]
    if lnfold:
        gamma, beta = 1.0 + 0.3 * torch.randn(K, generator=g), 0.2 * torch.randn(K, generator=g)
        ref = F.linear(F.layer_norm(x, (K,), gamma, beta, 1e-5), bf(w), bias)
        wf = bf(w * gamma[None, :])
        pk = ops.PackedConv(wf, 0, geglu=geglu, bias=bias + bf(w) @ beta)
        c1 = ops.PackedConv(wf, 0, geglu=geglu, bias=wf.sum(dim=1)).bias
        stats = ops.layernorm_stats(xd, 1e-5)
    else:
        ref = F.linear(x, bf(w), bias)
        pk = ops.PackedConv(bf(w), 0, geglu=geglu, bias=bias)
        c1 = stats = None
    raw = torch.zeros((M, N), device="cuda", dtype=torch.bfloat16) if geglu and lnfold else None
    proj = ref
    if geglu:
        hh, gg = ref.chunk(2, dim=-1)
        ref = hh * F.gelu(gg)
    if res:
        ref = ref + r
    part = torch.zeros((M, N // 40, 2), device="cuda", dtype=torch.float32) if rowstats else None
    ops.conv_gemm(xd, pk, 1, M, 1, M, 1, y=yd, res=rd, ksplit=1, ln_stats=stats, ln_c1=c1, rowpart=part, raw=raw, x_ld=xs.stride(0))
    torch.cuda.synchronize()
    assert_close(yd, ref, rtol=1.5e-2, atol=2e-2 if lnfold else 1e-3, what="ws gemm " + case[0])
    if views:
        assert float(ys[:, ncols:].float().abs().max()) == 0.0, "wrote outside the output view"
    if raw is not None:
        F2 = N // 2
        perm = torch.tensor([(q // 32) * 16 + (q % 32) if (q % 32) < 16 else F2 + (q // 32) * 16 + (q % 32 - 16) for q in range(N)])
        assert_close(raw, proj[:, perm], rtol=1.5e-2, atol=2e-2, what="ws geglu raw stash")
    if rowstats:
        assert_close(part[:, :, 0].sum(1), ref.sum(1), rtol=2e-3, atol=3e-2, what="ws row sums")
        assert_close(part[:, :, 1].sum(1), (ref * ref).sum(1), rtol=2e-3, atol=3e-2, what="ws row sums of squares")


@pytest.mark.parametrize("M,K,N", [(16384, 320, 320), (8192, 1280, 640), (3000, 640, 1280), (49152, 320, 320), (49152, 1280, 640)])
def test_linear_emits_layernorm_row_partials(ops, M, K, N):
    """CF_ROWSTATS: the to_out / ff.net.2 / proj_in GEMMs emit (sum, sum^2) of every output row per 80-column wave span; the LayerNorm
    that follows finalises (mean, rstd) from them instead of reading the tensor."""
    g = torch.Generator().manual_seed(78)
    x = bf(torch.randn(M, K, generator=g))
    w = bf(torch.randn(N, K, generator=g) / math.sqrt(K))
    bias = torch.randn(N, generator=g)
    res = bf(torch.randn(M, N, generator=g) * 2.0 + 0.5)
    pk = ops.PackedConv(w, 0, bias=bias)
    spans = N // 40       # the narrowest span any kernel form writes (gemm_ws.hip: 48 + 32 columns); the spans a form does not write stay zero
    part = torch.zeros((M, spans, 2), device="cuda", dtype=torch.float32)
    y = ops.conv_gemm(x.to(torch.bfloat16).cuda(), pk, 1, M, 1, M, 1, res=res.to(torch.bfloat16).cuda(), ksplit=1, rowpart=part)
    torch.cuda.synchronize()
    ref = x @ w.t() + bias + res
    assert_close(y, ref, what="rowstats: output")
    assert_close(part[:, :, 0].sum(1), ref.sum(1), rtol=2e-3, atol=2e-2, what="row sums")
    assert_close(part[:, :, 1].sum(1), (ref * ref).sum(1), rtol=2e-3, atol=2e-2, what="row sums of squares")
    stats = ops.layernorm_stats(y, 1e-5, rowpart=part, spans=spans)
    assert_close(stats[:, 0], ref.mean(1), rtol=2e-3, atol=2e-3, what="mean from partials")
    assert_close(stats[:, 1], (ref.var(1, unbiased=False) + 1e-5).rsqrt(), rtol=4e-3, atol=1e-4, what="rstd from partials")


def test_layernorm_row_partials_with_rows_far_from_zero(ops):
    """CF_ROWSTATS uses one-pass fp32 (sum, sum^2) of the fp32 values in front of their bf16 rounding; the folded GEMM then multiplies
    the bf16-stored rows.  With |row mean| = 50 x the row's standard deviation (beyond it bf16 storage itself -- quantum |mean| / 256 --
    drowns the row's variation) the one-pass variance must still give the rstd of the STORED rows to 4 % (measured 2.5 %: the statistics are
    those of the fp32 values, the stored rows carry the bf16 rounding noise -- quantum 0.5 at |v| = 70 -- on top), the mean to 1e-3."""
    g = torch.Generator().manual_seed(79)
    M, K, N = 16384, 320, 320
    x = bf(torch.randn(M, K, generator=g))
    w = bf(torch.randn(N, K, generator=g) / math.sqrt(K))
    res = bf(torch.randn(M, N, generator=g) + (torch.randint(0, 2, (M, 1), generator=g) * 2 - 1) * 70.0)
    pk = ops.PackedConv(w, 0)
    spans = N // 80
    part = torch.zeros((M, spans, 2), device="cuda", dtype=torch.float32)
    y = ops.conv_gemm(x.to(torch.bfloat16).cuda(), pk, 1, M, 1, M, 1, res=res.to(torch.bfloat16).cuda(), ksplit=1, rowpart=part)
    stats = ops.layernorm_stats(y, 1e-5, rowpart=part, spans=spans)
    torch.cuda.synchronize()
    ys = y.float().cpu()                                  # the stored (bf16-rounded) rows the consumer normalises
    ratio = (ys.mean(1).abs() / ys.std(1)).median()
    assert 40 < float(ratio) < 60, float(ratio)
    assert_close(stats[:, 0], ys.mean(1), rtol=1e-3, atol=1e-3, what="mean from partials, shifted rows")
    assert_close(stats[:, 1], (ys.var(1, unbiased=False) + 1e-5).rsqrt(), rtol=4e-2, atol=1e-4, what="rstd from partials, shifted rows")


ATT_CASES = [("self_d40", 2, 8, 256, 256, 40), ("self_d64", 1, 2, 200, 200, 64), ("self_d80", 1, 4, 128, 128, 80),
             ("self_d160", 1, 2, 64, 64, 160), ("cross77_d40", 2, 8, 256, 77, 40), ("cross77_d160", 1, 8, 64, 77, 160),
             ("self_d32", 1, 2, 96, 96, 32), ("vae_d512", 1, 1, 256, 256, 512)]


@pytest.mark.parametrize("case", ATT_CASES, ids=[c[0] for c in ATT_CASES])
def test_attention(ops, case):
    name, B, H, Nq, Nk, D = case
    g = torch.Generator().manual_seed(7)
    q = bf(torch.randn(B, Nq, H, D, generator=g))
    k = bf(torch.randn(B, Nk, H, D, generator=g))
    v = bf(torch.randn(B, Nk, H, D, generator=g))
    # a spiked key row forces the online-softmax rescale path
    k[0, Nk // 2, 0] *= 6.0
    k = bf(k)
    scale = 1.0 / math.sqrt(D)
    qr, kr, vr = (t.clone().requires_grad_(True) for t in (q, k, v))
    s = torch.einsum("bqhd,bkhd->bhqk", qr, kr) * scale
    ref = torch.einsum("bhqk,bkhd->bqhd", s.softmax(-1), vr)
    d_o = bf(torch.randn(B, Nq, H, D, generator=g))
    gq, gk, gv = torch.autograd.grad(ref, (qr, kr, vr), d_o)
    dev = lambda t, n: t.reshape(B * n, H * D).to(torch.bfloat16).cuda()
    cross = Nk == 77
    out = ops.attention(dev(q, Nq), dev(k, Nk), dev(v, Nk), B, H, Nq, Nk, D, scale, d_o=dev(d_o, Nq), need_dkv=not cross)
    o, lse, dq, dk, dv = out
    torch.cuda.synchronize()
    assert_close(o.reshape(B, Nq, H, D), ref.detach(), rtol=2e-2, atol=2e-3, what=name + " O")
    lse_ref = torch.logsumexp(s.detach(), dim=-1)
    assert_close(lse, lse_ref, rtol=1e-3, atol=1e-3, what=name + " LSE")
    assert_close(dq.reshape(B, Nq, H, D), gq, rtol=3e-2, atol=3e-3, what=name + " dQ")
    if not cross:
        assert_close(dk.reshape(B, Nk, H, D), gk, rtol=3e-2, atol=3e-3, what=name + " dK")
        assert_close(dv.reshape(B, Nk, H, D), gv, rtol=3e-2, atol=3e-3, what=name + " dV")


SHORTK_CASES = [("cross77_d40_4096q", 1, 8, 4096, 77, 40, False), ("cross77_d40_prescaled", 2, 8, 1024, 77, 40, True), ("cross77_d80", 2, 8, 1024, 77, 80, True),
                ("cross77_d64_sdxl", 1, 10, 4096, 77, 64, True), ("cross80_d40_full_last_tile", 1, 4, 512, 80, 40, False),
                ("cross1_d40_single_key", 1, 2, 64, 1, 40, False), ("cross65_d80", 1, 4, 96, 65, 80, False), ("cross16_d64_one_tile", 2, 2, 32, 16, 64, True),
                ("cross77_d40_many_tiles_per_wave", 8, 8, 4096, 77, 40, True)]


@pytest.mark.parametrize("case", SHORTK_CASES, ids=[c[0] for c in SHORTK_CASES])
def test_attention_short_keys(ops, case):
    """attention_shortk.hip: <= 80 keys (the 77-token cross-attention), K / V resident per workgroup, every wave walking its own 32-query
    tiles with the next tile's Q rows requested by LDS-DMA behind hand-counted waits; exact one-pass softmax.  O, LSE and the dQ the
    streaming backward derives from that LSE against torch; plain and prescaled queries; 1 .. 80 keys; one to eight tiles per wave; strided
    (column-view) Q / O rows as the engine passes them."""
    name, B, H, Nq, Nk, D, pre = case
    g = torch.Generator().manual_seed(len(name))
    scale = 1.0 / math.sqrt(D)
    q = bf(torch.randn(B, Nq, H, D, generator=g) * 1.3)
    k = bf(torch.randn(B, Nk, H, D, generator=g) * 1.3)
    v = bf(torch.randn(B, Nk, H, D, generator=g))
    if Nk > 2:
        k[0, Nk // 2, 0] *= 5.0            # a spiked key
        k = bf(k)
    qs = bf(q * (scale * 1.4426950408889634)) if pre else q          # what the engine's to_q produces when the scale is folded in
    qr, kr, vr = (t.clone().requires_grad_(True) for t in (qs if pre else q, k, v))
    s = torch.einsum("bqhd,bkhd->bhqk", qr, kr) * (0.6931471805599453 if pre else scale)
    ref = torch.einsum("bhqk,bkhd->bqhd", s.softmax(-1), vr)
    d_o = bf(torch.randn(B, Nq, H, D, generator=g))
    (gq,) = torch.autograd.grad(ref, (qr,), d_o)
    # Q lives inside a wider row (fused projections are consumed through column views)
    qw = torch.zeros((B * Nq, H * D + 64), dtype=torch.bfloat16, device="cuda")
    qw[:, 64:] = (qs if pre else q).reshape(B * Nq, H * D).to(torch.bfloat16).cuda()
    dev = lambda t, n: t.reshape(B * n, H * D).to(torch.bfloat16).cuda()
    o, lse, dq, _, _ = ops.attention(qw[:, 64:], dev(k, Nk), dev(v, Nk), B, H, Nq, Nk, D, 0.6931471805599453 if pre else scale,
                                     d_o=dev(d_o, Nq), need_dkv=False, q_prescaled=pre)
    torch.cuda.synchronize()
    assert_close(o.reshape(B, Nq, H, D), ref.detach(), rtol=2e-2, atol=2e-3, what=name + " O")
    assert_close(lse, torch.logsumexp(s.detach(), dim=-1), rtol=1e-3, atol=1e-3, what=name + " LSE")
    assert_close(dq.reshape(B, Nq, H, D), gq, rtol=3e-2, atol=3e-3, what=name + " dQ")


def test_short_key_kernel_is_as_accurate_as_the_streaming_kernel(ops):
    """The default cross-attention kernel (attention_shortk.hip, d = 40 with the row sums from a ones column) against the streaming forward
    it replaced (AttnParams.no_shortk) ON THE SAME INPUTS, both against fp32 softmax: O rel-L2 and the LSE error of the short-key kernel must
    not exceed the streaming kernel's by more than 5 % -- so the looser own-forward bound of tests/test_fullsize_gpu.py (conditioning of the
    guide's masks) cannot hide an accuracy regression of the kernel itself."""
    B, H, Nq, Nk, D = 2, 8, 4096, 77, 40
    g = torch.Generator().manual_seed(5)
    for gain in (1.0, 2.0):
        q = bf(torch.randn(B, Nq, H, D, generator=g) * gain * (1.4426950408889634 / math.sqrt(D)))       # prescaled, as the engine's to_q emits it
        k = bf(torch.randn(B, Nk, H, D, generator=g) * gain)
        v = bf(torch.randn(B, Nk, H, D, generator=g))
        s = torch.einsum("bqhd,bkhd->bhqk", q, k) * 0.6931471805599453
        ref = torch.einsum("bhqk,bkhd->bqhd", s.softmax(-1), v)
        ref_lse = torch.logsumexp(s, dim=-1)
        dev = lambda t, n: t.reshape(B * n, H * D).to(torch.bfloat16).cuda()
        err = {}
        for name, off in (("shortk", False), ("stream", True)):
            o, lse = ops.attention(dev(q, Nq), dev(k, Nk), dev(v, Nk), B, H, Nq, Nk, D, 0.6931471805599453, q_prescaled=True, no_shortk=off)
            torch.cuda.synchronize()
            e = o.float().cpu().reshape(B, Nq, H, D) - ref
            err[name] = (float(e.norm() / ref.norm()), float((lse.cpu() - ref_lse).abs().max()))
        print("cross-attention 77 keys d=40 gain %.0f: O rel-L2 short-key %.5f streaming %.5f | LSE max err %.5f / %.5f"
              % (gain, err["shortk"][0], err["stream"][0], err["shortk"][1], err["stream"][1]))
        assert err["shortk"][0] != err["stream"][0], "AttnParams.no_shortk did not select another kernel"
        assert err["shortk"][0] <= 1.05 * err["stream"][0] and err["shortk"][1] <= 1.05 * err["stream"][1] + 1e-4, err


PRE_CASES = [("self_d40", 2, 8, 512, 512, 40, 1.0), ("self_d40_peaky", 1, 8, 1024, 2048, 40, 2.0), ("cross77_d40", 2, 8, 256, 77, 40, 1.0),
             ("self_d64", 1, 2, 200, 200, 64, 1.5), ("self_d80", 1, 4, 256, 320, 80, 1.5), ("self_d160", 1, 2, 64, 64, 160, 1.0),
             # fewer keys than one 64-key tile (the tiny config's 4x4 / 8x8 levels and its 13-token prompt): the first tile is also the ragged one
             ("cross13_d32", 2, 2, 256, 13, 32, 1.0), ("self16_d64", 2, 2, 16, 16, 64, 1.0), ("self64_d32", 1, 2, 64, 64, 32, 1.0),
             ("cross13_d64", 2, 2, 64, 13, 64, 1.0)]


@pytest.mark.parametrize("case", PRE_CASES, ids=[c[0] for c in PRE_CASES])
def test_attention_with_prescaled_query(ops, case):
    """q_prescaled (AttnParams): the engine folds 1/sqrt(d) * log2(e) into the to_q weights, so the query tensor the kernels see is
    q' = c q and the scores q'.k are log2-domain; the forward (lazy-reference LDS-DMA kernel at d <= 80, the register-staged kernel at
    d = 160) and the backward get scale = ln 2.  Reference: softmax(ln 2 * q'.k) on the same bf16 q'.  Peaky case: scores of sigma 4
    plus a drift along the keys (tiles far above the first tile's reference are rebased)."""
    name, B, H, Nq, Nk, D, amp = case
    g = torch.Generator().manual_seed(11)
    c = math.log2(math.e) / math.sqrt(D)
    q = torch.randn(B, Nq, H, D, generator=g) * amp
    k = torch.randn(B, Nk, H, D, generator=g) * amp
    if amp > 1.9:
        u = torch.randn(H, D, generator=g)
        u = u / u.norm(dim=-1, keepdim=True)
        a = math.sqrt(12.0 * math.sqrt(D))
        q = q + a * u
        k = k + (torch.linspace(0, 1, Nk)[None, :, None, None] * a) * u
    qp, k = bf(q * c), bf(k)
    v = bf(torch.randn(B, Nk, H, D, generator=g))
    ln2 = math.log(2.0)
    qr, kr, vr = (t.clone().requires_grad_(True) for t in (qp, k, v))
    s = torch.einsum("bqhd,bkhd->bhqk", qr, kr) * ln2
    ref = torch.einsum("bhqk,bkhd->bqhd", s.softmax(-1), vr)
    d_o = bf(torch.randn(B, Nq, H, D, generator=g))
    gq, gk, gv = torch.autograd.grad(ref, (qr, kr, vr), d_o)
    dev = lambda t, n: t.reshape(B * n, H * D).to(torch.bfloat16).cuda()
    cross = Nk in (77, 13)
    o, lse, dq, dk, dv = ops.attention(dev(qp, Nq), dev(k, Nk), dev(v, Nk), B, H, Nq, Nk, D, ln2, d_o=dev(d_o, Nq), need_dkv=not cross,
                                       q_prescaled=True)
    torch.cuda.synchronize()
    assert_close(o.reshape(B, Nq, H, D), ref.detach(), rtol=2e-2, atol=2e-3, what=name + " O")
    assert_close(lse, torch.logsumexp(s.detach(), dim=-1), rtol=1e-3, atol=2e-3, what=name + " LSE")
    assert_close(dq.reshape(B, Nq, H, D), gq, rtol=3e-2, atol=3e-3, what=name + " dQ")
    if not cross:
        assert_close(dk.reshape(B, Nk, H, D), gk, rtol=3e-2, atol=3e-3, what=name + " dK")
        assert_close(dv.reshape(B, Nk, H, D), gv, rtol=3e-2, atol=3e-3, what=name + " dV")


@pytest.mark.parametrize("D,H,Nq", [(40, 8, 512), (80, 4, 192), (64, 2, 128)])
def test_attention_reference_runaway_takes_the_safe_sweep(ops, D, H, Nq):
    """The lazy forward's optimistic sweep takes no row maximum after the first key tile; rows whose later scores run away from the
    first tile's maximum by more than 2^64 (here: up to +150 in the log2 domain, far beyond anything a trained attention produces) must
    be caught by the row-sum check and recomputed by the safe sweep (per-tile maximum + rebase) -- same results as fp32 softmax."""
    B, Nk = 1, 1024 + 40                                  # ragged last tile as well
    g = torch.Generator().manual_seed(5)
    u = torch.randn(H, D, generator=g)
    u = u / u.norm(dim=-1, keepdim=True)
    a = math.sqrt(150.0)
    qp = bf(torch.randn(B, Nq, H, D, generator=g) * 0.3 + a * u)
    k = bf(torch.randn(B, Nk, H, D, generator=g) + (torch.linspace(0, 1, Nk)[None, :, None, None] * a) * u)
    v = bf(torch.randn(B, Nk, H, D, generator=g))
    ln2 = math.log(2.0)
    s = torch.einsum("bqhd,bkhd->bhqk", qp, k) * ln2
    assert float(s.max() - s[..., :64].max()) > 64 * ln2       # the run-away the optimistic sweep cannot represent
    ref = torch.einsum("bhqk,bkhd->bqhd", s.softmax(-1), v)
    dev = lambda t, n: t.reshape(B * n, H * D).to(torch.bfloat16).cuda()
    o, lse = ops.attention(dev(qp, Nq), dev(k, Nk), dev(v, Nk), B, H, Nq, Nk, D, ln2, q_prescaled=True)
    torch.cuda.synchronize()
    assert_close(o.reshape(B, Nq, H, D), ref, rtol=2e-2, atol=2e-3, what="runaway O")
    assert_close(lse, torch.logsumexp(s, dim=-1), rtol=1e-3, atol=2e-3, what="runaway LSE")


@pytest.mark.parametrize("D,H", [(40, 8), (80, 8)])
def test_attention_4096_keys_peaky_logits(ops, D, H):
    """The UNet's 64x64 self-attention (4096 queries x 4096 keys, d = 40; d = 80 at 32x32 uses the same kernel) with PEAKY logits:
    scaled scores of standard deviation >= 4 and row maxima that keep rising along the key axis, so the online softmax rescales in
    most of its 64 key tiles (synthetic fan_in^-1/2 weights give near-uniform scores, which never exercise that path at full length)."""
    B, Nq, Nk = 1, 4096, 4096
    g = torch.Generator().manual_seed(41)
    scale = 1.0 / math.sqrt(D)
    amp = 2.0                                               # q.k / sqrt(D) of N(0, amp^2) entries has sigma amp^2 = 4
    q = bf(torch.randn(B, Nq, H, D, generator=g) * amp)
    k = bf(torch.randn(B, Nk, H, D, generator=g) * amp)
    # a drift along the key axis (up to +12 in the scaled score): later keys align more with a direction every query shares, so the
    # running maxima keep moving tile after tile
    u = torch.randn(H, D, generator=g)
    u = u / u.norm(dim=-1, keepdim=True)
    a = math.sqrt(12.0 * math.sqrt(D))
    q = bf(q + a * u)
    k = bf(k + (torch.linspace(0, 1, Nk)[None, :, None, None] * a) * u)
    v = bf(torch.randn(B, Nk, H, D, generator=g))
    s = torch.einsum("bqhd,bkhd->bhqk", q, k) * scale
    assert float(s.std()) >= 4.0, float(s.std())
    ref = torch.einsum("bhqk,bkhd->bqhd", s.softmax(-1), v)
    dev = lambda t, n: t.reshape(B * n, H * D).to(torch.bfloat16).cuda()
    o, lse = ops.attention(dev(q, Nq), dev(k, Nk), dev(v, Nk), B, H, Nq, Nk, D, scale)
    torch.cuda.synchronize()
    assert_close(o.reshape(B, Nq, H, D), ref, rtol=2e-2, atol=2e-3, what="peaky O")
    assert_close(lse, torch.logsumexp(s, dim=-1), rtol=1e-3, atol=2e-3, what="peaky LSE")
    # backward on the same scores: dQ, dK, dV against autograd on a 512-query slice (the full 4096^2 x 8 autograd graph is 2 GB)
    Ns = 512
    qr, kr, vr = (t.clone().requires_grad_(True) for t in (q[:, :Ns], k, v))
    ss = torch.einsum("bqhd,bkhd->bhqk", qr, kr) * scale
    rs = torch.einsum("bhqk,bkhd->bqhd", ss.softmax(-1), vr)
    d_o = bf(torch.randn(B, Ns, H, D, generator=g))
    gq, gk, gv = torch.autograd.grad(rs, (qr, kr, vr), d_o)
    o2, lse2, dq, dk, dv = ops.attention(dev(q[:, :Ns], Ns), dev(k, Nk), dev(v, Nk), B, H, Ns, Nk, D, scale, d_o=dev(d_o, Ns))
    torch.cuda.synchronize()
    assert_close(dq.reshape(B, Ns, H, D), gq, rtol=3e-2, atol=3e-3, what="peaky dQ")
    assert_close(dk.reshape(B, Nk, H, D), gk, rtol=3e-2, atol=3e-3, what="peaky dK")
    assert_close(dv.reshape(B, Nk, H, D), gv, rtol=3e-2, atol=3e-3, what="peaky dV")


@pytest.mark.parametrize("case", [("self_1024", 2, 5, 1024, 1024, 1.0), ("self_4096_peaky", 1, 4, 1024, 4096, 2.5), ("cross_77", 2, 10, 512, 77, 1.0),
                                  ("ragged_300x200", 1, 2, 300, 200, 1.0)], ids=lambda c: c[0])
def test_attention_fp8_pv_d64(ops, case):
    """BASELINE.json configs[4]: "fp8 MFMA attention" -- the d = 64 forward with P.V on v_mfma_scale_f32_16x16x128_f8f6f4 (e4m3
    probabilities x e4m3 values, AttnParams.pv_fp8; QK^T, softmax, row sums and LSE unchanged) against fp32 softmax attention on the same
    bf16 inputs.  STATED TOLERANCE: e4m3 carries 3 mantissa bits (relative rounding error up to 2^-4 per probability and per value,
    independent per key): relative L2 of O <= 5 % (measured 3.2 - 3.8 %: 0.2 % for the bf16 kernel on the same inputs), every element
    within 2^-4 x 1.1 of the largest value magnitude, LSE as the bf16 kernel (it does not pass through fp8).  Flat scores (the output is a
    small average of many values: the absolute error is 0.002 rms), peaky scores (sigma ~ 6: few keys per row carry the mass, nothing
    averages down), the 77-key cross-attention (one partial 128-key tile) and ragged sizes."""
    name, B, H, Nq, Nk, gain = case
    D = 64
    g = torch.Generator().manual_seed(29)
    q = bf(torch.randn(B, Nq, H, D, generator=g) * gain)
    k = bf(torch.randn(B, Nk, H, D, generator=g) * gain)
    v = bf(torch.randn(B, Nk, H, D, generator=g))
    scale = 1.0 / math.sqrt(D)
    s = torch.einsum("bqhd,bkhd->bhqk", q, k) * scale
    ref = torch.einsum("bhqk,bkhd->bqhd", s.softmax(-1), v)
    dev = lambda t, n: t.reshape(B * n, H * D).to(torch.bfloat16).cuda()
    o8, lse8 = ops.attention(dev(q, Nq), dev(k, Nk), dev(v, Nk), B, H, Nq, Nk, D, scale, pv_fp8=True)
    o16, lse16 = ops.attention(dev(q, Nq), dev(k, Nk), dev(v, Nk), B, H, Nq, Nk, D, scale)
    torch.cuda.synchronize()
    r8 = float((o8.float().cpu().reshape(B, Nq, H, D) - ref).norm() / ref.norm())
    r16 = float((o16.float().cpu().reshape(B, Nq, H, D) - ref).norm() / ref.norm())
    m8 = float((o8.float().cpu().reshape(B, Nq, H, D) - ref).abs().max())
    print("%s: O rel-L2 vs fp32  fp8 P.V %.4f (max abs %.4f)   bf16 P.V %.4f   score sigma %.1f" % (name, r8, m8, r16, float(s.std())))
    assert torch.isfinite(o8.float()).all()
    assert r8 <= 0.05 and m8 <= 1.1 / 16 * float(v.abs().max()), (r8, m8)
    assert r16 <= 0.01
    # LSE does not pass through fp8.  The fp8 kernel is the lazy-reference form: queries the producer has not prescaled are multiplied by
    # scale * log2(e) and rounded to bf16 once more (2^-9 of the score magnitude: 0.05 at the peaky case's scores of 37; inside the
    # engine the projection GEMM prescales and both kernels see the same bits)
    assert_close(lse8, lse16, rtol=2e-3, atol=1e-3, what=name + " LSE (fp8 P.V against the bf16 kernel)")
    assert_close(lse8, torch.logsumexp(s, dim=-1), rtol=2e-3, atol=2e-3, what=name + " LSE (fp8 P.V)")
    assert not torch.equal(o8, o16), "the fp8 path did not run"


@pytest.mark.parametrize("case", [("vae_mid_d512", 2, 1, 1024, 1024, 512, 8), ("wide_2heads_d256", 1, 2, 512, 768, 256, 8),
                                  ("vae_mid_d512_groups_2_2_1", 5, 1, 1024, 1024, 512, 2), ("vae_mid_d512_one_image_scratch", 3, 1, 512, 768, 512, 1),
                                  ("vae_mid_d512_grouped_launches", 8, 1, 2048, 2048, 512, 8)], ids=lambda c: c[0])
def test_attention_gemm_path(ops, case):
    """wide heads through the GEMM kernel (attention_gemm.hip): same contract and tolerances as the flash kernels.  Single-head layers run
    several images per launch (grouped GEMMs: one weight matrix per 256-row-aligned row group of the persistent kernel, batched
    transposes / softmax): a full group, ragged groups (5 images in groups of 2), one image of scratch, a 768-key case that is not a
    multiple of 256 rows (stays per image); the 8 x 2048-token case is large enough that every product really is ONE grouped launch
    (smaller groups fall back to one GEMM per image when the persistent kernel would not fill the chip)."""
    name, B, H, Nq, Nk, D, ws_images = case
    g = torch.Generator().manual_seed(23)
    q = bf(torch.randn(B, Nq, H, D, generator=g))
    k = bf(torch.randn(B, Nk, H, D, generator=g))
    v = bf(torch.randn(B, Nk, H, D, generator=g))
    scale = 1.0 / math.sqrt(D)
    qr, kr, vr = (t.clone().requires_grad_(True) for t in (q, k, v))
    s = torch.einsum("bqhd,bkhd->bhqk", qr, kr) * scale
    ref = torch.einsum("bhqk,bkhd->bqhd", s.softmax(-1), vr)
    d_o = bf(torch.randn(B, Nq, H, D, generator=g))
    gq, gk, gv = torch.autograd.grad(ref, (qr, kr, vr), d_o)
    dev = lambda t, n: t.reshape(B * n, H * D).to(torch.bfloat16).cuda()
    o, lse, dq, dk, dv = ops.attention_gemm(dev(q, Nq), dev(k, Nk), dev(v, Nk), B, H, Nq, Nk, D, scale, d_o=dev(d_o, Nq), ws_images=ws_images)
    torch.cuda.synchronize()
    assert_close(o.reshape(B, Nq, H, D), ref.detach(), rtol=2e-2, atol=2e-3, what=name + " O")
    assert_close(lse, torch.logsumexp(s.detach(), dim=-1), rtol=1e-3, atol=1e-3, what=name + " LSE")
    assert_close(dq.reshape(B, Nq, H, D), gq, rtol=3e-2, atol=3e-3, what=name + " dQ")
    assert_close(dk.reshape(B, Nk, H, D), gk, rtol=3e-2, atol=3e-3, what=name + " dK")
    assert_close(dv.reshape(B, Nk, H, D), gv, rtol=3e-2, atol=3e-3, what=name + " dV")


@pytest.mark.parametrize("case", [("clip_causal_77_d64", 3, 12, 77, 77, 64), ("causal_130_d64", 2, 2, 130, 130, 64)], ids=lambda c: c[0])
def test_attention_causal(ops, case):
    """causal forward (CLIP text encoder: key j visible to query i iff j <= i)."""
    name, B, H, Nq, Nk, D = case
    g = torch.Generator().manual_seed(17)
    q = bf(torch.randn(B, Nq, H, D, generator=g))
    k = bf(torch.randn(B, Nk, H, D, generator=g))
    v = bf(torch.randn(B, Nk, H, D, generator=g))
    scale = 1.0 / math.sqrt(D)
    s = torch.einsum("bqhd,bkhd->bhqk", q, k) * scale
    s = s.masked_fill(torch.ones(Nq, Nk).triu(1).bool(), float("-inf"))
    ref = torch.einsum("bhqk,bkhd->bqhd", s.softmax(-1), v)
    dev = lambda t, n: t.reshape(B * n, H * D).to(torch.bfloat16).cuda()
    o, lse = ops.attention(dev(q, Nq), dev(k, Nk), dev(v, Nk), B, H, Nq, Nk, D, scale, causal=True)
    torch.cuda.synchronize()
    assert_close(o.reshape(B, Nq, H, D), ref, rtol=2e-2, atol=2e-3, what=name + " O")
    assert_close(lse, torch.logsumexp(s, dim=-1), rtol=1e-3, atol=1e-3, what=name + " LSE")


def test_elementwise_sampler_ops(ops):
    from distdiff_amd import _lib
    L = _lib.lib()
    g = torch.Generator().manual_seed(8)
    B, Cc, H, W = 2, 4, 8, 8
    HW = H * W
    P = lambda t: C.c_void_p(t.data_ptr())
    # CFG + DDIM step and its VJP
    z = torch.randn(B, Cc, H, W, generator=g)
    eps2 = torch.randn(2 * B, Cc, H, W, generator=g)
    coef = torch.tensor([7.5, 0.8, 0.6, 0.9, 0.43589])
    zr = z.clone().requires_grad_(True)
    er = eps2.clone().requires_grad_(True)
    eu, ec = er.chunk(2)
    eps = eu + coef[0] * (ec - eu)
    x0 = (zr - coef[2] * eps) / coef[1]
    zp = coef[3] * x0 + coef[4] * eps
    eps_nhwc = torch.zeros(2 * B * HW, 8)
    eps_nhwc[:, :Cc] = eps2.permute(0, 2, 3, 1).reshape(-1, Cc)
    d_eps, d_z, d_coef = eps_nhwc.cuda(), z.cuda(), coef.cuda()
    d_zp, d_x0 = torch.empty_like(d_z), torch.empty_like(d_z)
    _lib.check(L.dd_op_cfg_ddim(P(d_eps), 8, P(d_z), P(d_zp), P(d_x0), B, Cc, HW, P(d_coef), None))
    assert_close(d_zp, zp.detach(), rtol=1e-5, atol=1e-5, what="ddim z_prev")
    assert_close(d_x0, x0.detach(), rtol=1e-5, atol=1e-5, what="ddim x0")
    gx0, gzp = torch.randn(B, Cc, H, W, generator=g), torch.randn(B, Cc, H, W, generator=g)
    gz_ref, ge_ref = torch.autograd.grad([x0, zp], [zr, er], [gx0, gzp])
    d_ge = torch.zeros((2 * B * HW, 8), device="cuda", dtype=torch.bfloat16)
    d_gz = torch.empty_like(d_z)
    d_gx0, d_gzp = gx0.cuda(), gzp.cuda()   # keep references alive: the kernels are asynchronous
    _lib.check(L.dd_op_cfg_ddim_bwd(P(d_gx0), P(d_gzp), P(d_ge), 8, P(d_gz), B, Cc, HW, P(d_coef), None))
    assert_close(d_gz, gz_ref, rtol=1e-5, atol=1e-5, what="ddim bwd g_z")
    assert_close(d_ge.float().cpu()[:, :Cc].reshape(2 * B, H, W, Cc).permute(0, 3, 1, 2), ge_ref, rtol=1e-2, what="ddim bwd g_eps")
    # transform-guidance update
    e, b = torch.rand(B * Cc, generator=g), torch.randn(B * Cc, generator=g)
    gg = torch.randn(B, Cc, H, W, generator=g) * 0.01
    rho, cval = 10.0, 0.2
    ge = (gg * z).sum((2, 3)).reshape(-1)
    gb = gg.sum((2, 3)).reshape(-1)
    en, bn = e - rho * ge, b - rho * gb
    znew = z * (1 + en.reshape(B, Cc, 1, 1)) + bn.reshape(B, Cc, 1, 1)
    znew = torch.max(torch.min(znew, z + cval), z - cval)
    d_out = torch.empty_like(d_z)
    d_gg, d_e, d_b = gg.cuda(), e.cuda(), b.cuda()
    _lib.check(L.dd_op_transform_update(P(d_z), P(d_gg), P(d_e), P(d_b), P(d_out), B * Cc, HW, rho, cval, None))
    assert_close(d_out, znew, rtol=1e-5, atol=1e-5, what="transform update")


def test_bicubic_maxpool_gap_energy(ops):
    from distdiff_amd import _lib
    L = _lib.lib()
    P = lambda t: C.c_void_p(t.data_ptr())
    g = torch.Generator().manual_seed(9)
    B, Hs, Hd = 2, 64, 28   # same 16/7 ratio as 512 -> 224
    img = bf(torch.randn(B, 3, Hs, Hs, generator=g))
    ir = img.clone().requires_grad_(True)
    ref = F.interpolate(ir, size=(Hd, Hd), mode="bicubic")
    src = ops.to_nhwc_bf16(img, 8).cuda()
    dst = torch.empty((B * Hd * Hd, 8), device="cuda", dtype=torch.bfloat16)
    _lib.check(L.dd_op_bicubic(P(src), 8, P(dst), 8, B, Hs, Hs, Hd, Hd, 3, 8, None))
    assert_close(ops.from_nhwc(dst, B, Hd, Hd)[:, :3], ref.detach(), what="bicubic fwd")
    assert float(dst[:, 3:].float().abs().max()) == 0.0
    dd = bf(torch.randn(B, 3, Hd, Hd, generator=g))
    (gi,) = torch.autograd.grad(ref, ir, dd)
    ddst = ops.to_nhwc_bf16(dd, 8).cuda()
    dsrc = torch.zeros((B * Hs * Hs, 8), device="cuda", dtype=torch.bfloat16)
    _lib.check(L.dd_op_bicubic_bwd(P(ddst), 8, P(dsrc), 8, B, Hs, Hs, Hd, Hd, 3, None))
    assert_close(ops.from_nhwc(dsrc, B, Hs, Hs)[:, :3], gi, what="bicubic bwd")
    # max pool 3x3/2 (+bwd)
    x = bf(torch.randn(B, 64, 16, 16, generator=g))
    xr = x.clone().requires_grad_(True)
    mp = F.max_pool2d(xr, 3, 2, 1)
    xd = ops.to_nhwc_bf16(x).cuda()
    yd = torch.empty((B * 8 * 8, 64), device="cuda", dtype=torch.bfloat16)
    _lib.check(L.dd_op_maxpool3x3s2(P(xd), P(yd), B, 16, 16, 64, None))
    assert_close(ops.from_nhwc(yd, B, 8, 8), mp.detach(), rtol=0, atol=0, what="maxpool")
    dy = bf(torch.randn(B, 64, 8, 8, generator=g))
    (gx,) = torch.autograd.grad(mp, xr, dy)
    dxd = torch.empty_like(xd)
    dyd = ops.to_nhwc_bf16(dy).cuda()
    _lib.check(L.dd_op_maxpool3x3s2_bwd(P(xd), P(dyd), P(dxd), B, 16, 16, 64, None))
    assert_close(ops.from_nhwc(dxd, B, 16, 16), gx, rtol=1e-2, what="maxpool bwd")
    # GAP + energy (+ gradient), both guidance flavours
    D, K, Ccls = 256, 3, 5
    feat = bf(torch.randn(B, 7 * 7, D, generator=g).abs())
    fd = torch.empty((B, D), device="cuda", dtype=torch.float32)
    featd = feat.reshape(-1, D).to(torch.bfloat16).cuda()
    _lib.check(L.dd_op_gap(P(featd), D, P(fd), B, 49, D, None))
    assert_close(fd, feat.mean(1), rtol=1e-5, atol=1e-5, what="gap")
    Pc = F.normalize(torch.randn(Ccls, D, generator=g), dim=-1)
    Pg = F.normalize(torch.randn(Ccls, K, D, generator=g), dim=-1)
    tg = torch.tensor([3, 1], dtype=torch.int32)
    for normalize in (0, 1):
        f = feat.mean(1).clone().requires_grad_(True)
        fh = f / f.norm(dim=-1, keepdim=True) if normalize else f
        gp = Pc[tg.long()]
        sc = torch.norm(fh - gp, dim=1, p=2).mean() * 1.0
        lp = Pg[tg.long()]
        idx = torch.argmax(torch.bmm(fh.unsqueeze(1), lp.permute(0, 2, 1)), -1)
        lp = lp[torch.arange(B), idx.squeeze()]
        sc = sc + torch.norm(fh - lp, dim=1, p=2).mean() * 0.7
        sc = sc * 0.5
        (gf_ref,) = torch.autograd.grad(sc, f)
        score = torch.zeros(1, device="cuda")
        gf = torch.empty((B, D), device="cuda")
        dPc, dPg, dtg = Pc.cuda(), Pg.cuda(), tg.cuda()
        _lib.check(L.dd_op_energy(P(fd), P(dPc), P(dPg), P(dtg), B, D, K, 1.0, 0.7, 1, 1, normalize, 0.5,
                                  P(score), P(gf), None))
        torch.cuda.synchronize()
        assert_close(score, sc.detach().reshape(1), rtol=1e-5, atol=1e-6, what="energy score n=%d" % normalize)
        assert_close(gf, gf_ref, rtol=1e-4, atol=1e-7, what="energy grad n=%d" % normalize)


STATS_CASES = [
    # name, B, Cin, Cout, H, W, k, res (exercise: 256x160 FE epilogue, 128x160 two-workgroup form, deep-K generic epilogue, 256x128)
    # (shapes large enough for the persistent kernel without split-K: >= 192 work items)
    ("fe_128x160_3x3", 2, 64, 320, 96, 96, 3, True),
    ("fe_1x1_res", 2, 320, 640, 64, 64, 1, True),
    ("generic_256x160_deepK", 2, 960, 320, 128, 128, 3, False),
    ("two_wg_128x128", 1, 128, 128, 256, 256, 3, True),
    ("halo_320_res", 48, 128, 320, 32, 32, 3, True),          # conv_halo.hip: 256 x 320 tiles, residual + partials
    ("halo_256", 12, 64, 256, 64, 64, 3, False),               # 256 x 256 tiles
    ("halo_w256_res", 1, 64, 256, 256, 256, 3, True),          # 2 x 128-pixel tiles: partial blocks of 64 rows inside a tile row
    ("halo_n128_res", 1, 64, 128, 512, 512, 3, True),          # 512 x 128 tiles
    # channels far from zero: every GroupNorm group's bias is shifted by +-50 standard deviations of the convolution's output, so the
    # one-pass (mean, M2) of a 64-row block is the small difference sq - sa * mean of two numbers 2500 x larger (fp32 partials, Chan merge)
    ("halo_320_res_shift50", 48, 128, 320, 32, 32, 3, True, 50.0),
    ("fe_1x1_res_shift50", 2, 320, 640, 64, 64, 1, True, 50.0),
    ("two_wg_128x128_shift50", 1, 128, 128, 256, 256, 3, True, 50.0),
]


@pytest.mark.parametrize("case", STATS_CASES, ids=[c[0] for c in STATS_CASES])
def test_conv_emits_groupnorm_partials(ops, case):
    """CF_STATS: the implicit-GEMM epilogue also emits per-(64-row block, channel) partial (mean, M2) of the values it stores, and
    GroupNorm(+SiLU) computed from those partials (no statistics pass over the tensor) matches torch on the conv output."""
    name, B, Cin, Cout, H, W, k, with_res = case[:8]
    shift = case[8] if len(case) > 8 else 0.0
    g = torch.Generator().manual_seed(len(name))
    x = bf(torch.randn(B, Cin, H, W, generator=g))
    w = bf(torch.randn(Cout, Cin, k, k, generator=g) / math.sqrt(Cin * k * k))
    bias = torch.randn(Cout, generator=g)
    if shift:                                             # +-shift per GroupNorm group (32 groups)
        bias = bias + shift * (torch.arange(Cout) // (Cout // 32) % 2 * 2 - 1).float()
    res = bf(torch.randn(B, Cout, H, W, generator=g)) if with_res else None
    ref = F.conv2d(x, w, bias, padding=k // 2) + (res if with_res else 0)
    M = B * H * W
    xd = ops.to_nhwc_bf16(x, Cin).cuda()
    pk = ops.PackedConv(w, k // 2, mode=0, bias=bias)
    # the tensor sits in columns [64, 64 + Cout) of a wider "concat" buffer: strided output, strided partials
    Ct = Cout + 64
    ybuf = torch.zeros(M, Ct, device="cuda", dtype=torch.bfloat16)
    part = torch.zeros(M // 64, Ct, 2, device="cuda", dtype=torch.float32)
    resd = ops.to_nhwc_bf16(res, Cout).cuda() if with_res else None
    y = ops.conv_gemm(xd, pk, B, H, W, H, W, res=resd, y=ybuf[:, 64:], stats=part[:, 64:])
    torch.cuda.synchronize()
    assert_close(ops.from_nhwc(ybuf[:, 64:], B, H, W), ref, what=name + " conv")
    rows = ref.permute(0, 2, 3, 1).reshape(M // 64, 64, Cout)              # the fp32 values before the bf16 rounding of the store
    pm, pM2 = part[:, 64:, 0].cpu(), part[:, 64:, 1].cpu()
    assert_close(pm, rows.mean(1), rtol=2e-3, atol=2e-3, what=name + " partial mean")
    # shifted: |mean| / std = 50 / sqrt(1 + res) -> the fp32 one-pass M2 of 64 values near 50 keeps ~1e-3 of 64 * var (measured), bound 5 %
    assert_close(pM2, ((rows - rows.mean(1, keepdim=True)) ** 2).sum(1), rtol=5e-2 if shift else 2e-2, atol=2e-2, what=name + " partial M2")
    assert float(part[:, :64].abs().max()) == 0.0                          # nothing outside the op's channel range
    # GroupNorm from the partials
    G, eps = 32, 1e-5
    gamma, beta = torch.randn(Cout, generator=g), torch.randn(Cout, generator=g)
    yv = ybuf[:, 64:]
    got, stats = ops.groupnorm(yv, gamma.cuda(), beta.cuda(), B, H * W, G, eps, True, chan_part=part[:, 64:])
    xq = ops.from_nhwc(yv, B, H, W).cpu()
    want = F.silu(F.group_norm(xq, G, gamma, beta, eps))
    assert_close(ops.from_nhwc(got, B, H, W), want, what=name + " gn from partials")
    plain, stats2 = ops.groupnorm(yv, gamma.cuda(), beta.cuda(), B, H * W, G, eps, True)
    assert_close(stats, stats2, rtol=2e-3, atol=2e-3, what=name + " stats vs statistics pass")


def test_halo_switch_fallbacks_stay_correct():
    """The A/B switches of the halo kernels select code that the default configuration never runs (they are read once per process):
    no address table (`DD_HALO_TAB=0`: per-piece address arithmetic, also what a geometry whose table does not fit in LDS gets), decoder
    levels on the one-tile form (`DD_HALO_PERSIST=0`), 4 x 128-pixel tiles (`DD_HALO_TW512=128`, whose 780-pixel halo leaves no room for
    the table in the 512 x 160 form).  The halo cases of this file again, in a child process with all three set."""
    import subprocess
    import sys
    env = dict(os.environ, DD_HALO_TAB="0", DD_HALO_PERSIST="0", DD_HALO_TW512="128")
    out = subprocess.run([sys.executable, "-m", "pytest", os.path.abspath(__file__), "-m", "gpu", "-q", "-x", "-k",
                          "test_conv_forward_and_dgrad and halo"], capture_output=True, text=True, timeout=900, env=env,
                         cwd=os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
    assert out.returncode == 0, out.stdout[-3000:] + out.stderr[-2000:]
    assert " passed" in out.stdout and "failed" not in out.stdout, out.stdout[-1000:]
