"""Launch groups (cgic_group_begin / _select / _launch): the shape groups of a tiled image through ONE launch per kernel.

The reference runs model.compress tile by tile (inference_high_resolution.py:236-257); grouping is a property of how the launches
are issued, never of the results: every test here compares the grouped chain with the ungrouped calls, byte for byte."""
import ctypes

import numpy as np
import pytest
import torch


def _setup(seed):
    import control_gic_amd as cg
    dev = torch.device("cuda", 0)
    rng = np.random.default_rng(seed)
    vq = cg.VectorQuantizer(1024, 4, beta=0.25).to(dev)
    with torch.no_grad():
        vq.embedding.weight.copy_(torch.from_numpy(rng.standard_normal((1024, 4)).astype(np.float32)))
    vq.usage_counter.copy_(torch.from_numpy(rng.integers(1, 1000, 1024).astype(np.float32)))
    codec = cg.GrainCodec(vq.embedding_counter, vq.embedding.weight.detach())
    return cg, dev, rng, vq, codec


def _same_decoded(a, b):
    for (i0, m0, z0), (i1, m1, z1) in zip(a, b):
        assert torch.equal(i0, i1) and torch.equal(z0, z1) and all(torch.equal(p, q) for p, q in zip(m0, m1))


def test_group_api_misuse_is_refused():
    """host logic only (no launch): begin twice, select out of range, launch / select without a group, too many groups"""
    from control_gic_amd import _lib
    l = _lib.lib()
    assert l.cgic_group_max() >= 4
    assert l.cgic_group_select(0) == _lib.ERR_INVALID
    assert l.cgic_group_launch(None) == _lib.ERR_INVALID
    assert l.cgic_group_begin(l.cgic_group_max() + 1, None) == _lib.ERR_INVALID
    assert l.cgic_group_begin(0, None) == _lib.ERR_INVALID
    bad = (ctypes.c_double * 2)(0.5, 0.0)
    assert l.cgic_group_begin(2, bad) == _lib.ERR_INVALID
    assert l.cgic_group_begin(2, None) == 0
    try:
        assert l.cgic_group_begin(2, None) == _lib.ERR_INVALID
        assert l.cgic_group_select(2) == _lib.ERR_INVALID
        assert l.cgic_group_select(1) == 0
        # entry points without a recorded form refuse to run inside a group (their launch would overtake the recorded ones)
        assert l.cgic_cut_tiles(None, 0, 1, 16, 16, 1, None, None) == _lib.ERR_INVALID
        assert b"no recorded form" in l.cgic_last_error()
        assert l.cgic_avgpool_f32(None, 1, 4, 4, 2, None, None) == _lib.ERR_INVALID and b"no recorded form" in l.cgic_last_error()
    finally:
        l.cgic_group_abort()
    assert l.cgic_group_select(0) == _lib.ERR_INVALID           # closed again


@pytest.mark.gpu
def test_an_exception_inside_the_block_closes_the_group():
    from control_gic_amd import _lib
    l = _lib.lib()
    with pytest.raises(RuntimeError):
        with _lib.launch_group(2, None, torch.device("cuda", 0)):
            raise RuntimeError("x")
    assert l.cgic_group_begin(1, None) == 0          # not left open
    l.cgic_group_abort()
    assert _lib.current_stream(torch.device("cuda", 0)) == torch.cuda.current_stream().cuda_stream


@pytest.mark.gpu
@pytest.mark.parametrize("H,W", [(1356, 2040), (1000, 1800), (700, 500), (768, 1000), (300, 900)])
@pytest.mark.parametrize("content", ["noise", "smooth8"])
def test_chain_of_ragged_tiles_equals_the_groups_one_by_one(H, W, content):
    """compress_tiled / decompress_tiled with chain=True: the (up to four) shape groups of one image as ONE launch per kernel ==
    the groups launched one after the other: every stream byte, bpp, decoded index, mask element and codebook row; three launches
    for the encode side, two for the decode side, whatever the number of groups.  smooth8: 8-bit gradients, i.e. bands with
    members -- the router's threshold-band refinement from the pixels runs inside the grouped launch too"""
    from control_gic_amd import highres
    from control_gic_amd.quantize import vq_forward_route
    cg, dev, rng, vq, codec = _setup(H * 7 + W)
    if content == "noise":
        x = torch.from_numpy(rng.random((1, 3, H, W), dtype=np.float32)).to(dev)
    else:
        yy, xx = np.mgrid[0:H, 0:W]
        g = ((yy * 0.11 + xx * 0.07) % 256).astype(np.uint8)
        x = torch.from_numpy((np.stack([g, g, (g // 2) * 2]) / 255.0).astype(np.float32)[None]).to(dev)
    from control_gic_amd.quantize import prepare_codebook
    prepared = prepare_codebook(vq.embedding.weight.detach())

    def encode(tiles):                      # a stand-in encoder that is a function of each tile's own pixels
        z = torch.nn.functional.avg_pool2d(tiles, 4)
        z = torch.cat([z, z[:, :1] * 2 - 1], dim=1) * 3 - 1.5
        e8, e16 = cg.entropy_maps(tiles)
        _, _, ind, mask, _, mode = vq_forward_route(z.contiguous(), vq.embedding.weight, 0.25, True, e16, e8, 0.1, 0.8, per_image=True,
                                                    pixels=tiles, prepared=prepared)
        return ind, mask, mode

    ref = highres.compress_tiled(x, encode, codec)
    got = highres.compress_tiled(x, encode, codec, chain=True)
    torch.cuda.synchronize()
    assert ref.tiles == got.tiles and ref.streams() == got.streams() and ref.bpp() == got.bpp()
    for (_, _, (i0, m0, _)), (_, _, (i1, m1, _)) in zip(ref.groups, got.groups):
        assert torch.equal(i0, i1) and all(torch.equal(p, q) for p, q in zip(m0, m1))
    for mode in ("latency", "throughput"):
        with cg.decoder_mode(mode):
            dref, _ = highres.decompress_tiled(ref, codec)
            dgot, _ = highres.decompress_tiled(got, codec, chain=True)
        _same_decoded(dref, dgot)


@pytest.mark.gpu
@pytest.mark.parametrize("frames,N,H,W", [(False, 1, 1356, 2040), (True, 1, 1356, 2040), (False, 3, 800, 1040), (False, 1, 512, 768)])
def test_one_call_tiled_driver_equals_the_group_by_group_path(frames, N, H, W):
    """cgic_compress_tiled (highres.TiledCall): the tiles of N images of one size through ONE foreign call -- cut + maps, VQ + router,
    coder, decoder + merge as one launch chain over buffers allocated once -- == highres.compress_tiled / decompress_tiled group by
    group (themselves pinned to the real reference's per-tile files): every stream byte, bpp, decoded index, mask and row; fp32
    images and uint8 frames, a 2040x1356 image (four shape groups), three 1040x800 ones, a single-shape 768x512 one; twice in a row"""
    from control_gic_amd import highres
    from control_gic_amd.quantize import vq_forward_route
    cg, dev, rng, vq, codec = _setup(7)
    tc = highres.TiledCall(vq, 0.1, 0.8, N, H, W, frequency=codec.huffman, frames=frames)
    zs = [torch.from_numpy(rng.standard_normal((N * len(idxs), 4, th // 4, tw // 4), dtype=np.float32)).to(dev) for (th, tw), idxs in tc.groups]
    zmap = {(N * len(idxs), th, tw): z for ((th, tw), idxs), z in zip(tc.groups, zs)}

    def encode(tiles):
        if frames:
            _, e8, e16 = cg.entropy_maps_u8(tiles, want_x=False)
            key = (tiles.shape[0], tiles.shape[1], tiles.shape[2])
        else:
            e8, e16 = cg.entropy_maps(tiles)
            key = (tiles.shape[0], tiles.shape[2], tiles.shape[3])
        _, _, ind, mask, _, mode = vq_forward_route(zmap[key], vq.embedding.weight, 0.25, True, e16, e8, 0.1, 0.8, per_image=True, pixels=tiles)
        return ind, mask, mode

    for rep in range(2):
        if rep == 1:        # smooth 8-bit content: long threshold bands, which the row bands of the full 768x768 tiles split between them (ws_refine)
            yy, xx = np.mgrid[0:H, 0:W].astype(np.float32)
            sm = np.stack([np.stack([np.round(255.0 * np.clip(0.5 + 0.45 * np.sin((0.6 * xx + (0.8 + 0.1 * n) * yy) * 0.7 * 2 * np.pi / W) * (0.7 + 0.1 * c)
                                                             + rng.integers(-1, 2, (H, W)) / 255.0, 0, 1)) for c in range(3)]) for n in range(N)]).astype(np.float32) / 255.0
            x = torch.from_numpy(np.ascontiguousarray((sm * 255.0).round().astype(np.uint8).transpose(0, 2, 3, 1)) if frames else sm).to(dev)
        elif frames:
            x = torch.from_numpy(rng.integers(0, 256, (N, H, W, 3), dtype=np.uint8)).to(dev)
        else:
            x = torch.from_numpy((rng.integers(0, 256, (N, 3, H, W)) / 255.0).astype(np.float32)).to(dev)
        got = tc(x, zs)
        torch.cuda.synchronize()
        ref = highres.compress_tiled_batch(x, encode, codec)
        got = [got] if N == 1 else got
        for g, r in zip(got, ref):
            assert g.tiles == r.tiles and g.streams() == r.streams() and g.bpp() == r.bpp()
        dref = highres.decompress_tiled_batch(ref, codec)
        for k, ((th, tw), idxs) in enumerate(tc.groups):
            dind, dmask, dzq, status = tc.decoded[k]
            assert int(status.abs().max()) == 0
            T = len(idxs)
            for n in range(N):
                for j, i in enumerate(idxs):
                    ind_r, masks_r, zq_r = dref[n][i]
                    b = n * T + j
                    assert torch.equal(dind[b:b + 1], ind_r) and torch.equal(dzq[b:b + 1], zq_r)
                    assert all(torch.equal(m[b:b + 1], q) for m, q in zip(dmask, masks_r))


@pytest.mark.gpu
def test_chain_launch_counts_and_uint8_frames():
    """a 2040x1356 image (6 tiles, 4 shape groups): the encode chain is 3 launches (2 + the tile cut inside the map launch), the decode chain 1; uint8 frames go through
    the grouped ToTensor + entropy launch"""
    from control_gic_amd import highres, _lib
    from control_gic_amd.quantize import vq_forward_route
    cg, dev, rng, vq, codec = _setup(5)
    H, W = 1356, 2040
    frames = torch.from_numpy(rng.integers(0, 256, (1, H, W, 3), dtype=np.uint8)).to(dev)
    zs = {}

    def encode_u8(tiles):
        T, th, tw, _ = tiles.shape
        if (T, th, tw) not in zs:
            zs[(T, th, tw)] = torch.from_numpy(np.random.default_rng(th + tw).standard_normal((T, 4, th // 4, tw // 4), dtype=np.float32)).to(dev)
        _, e8, e16 = cg.entropy_maps_u8(tiles, want_x=False)
        _, _, ind, mask, _, mode = vq_forward_route(zs[(T, th, tw)], vq.embedding.weight, 0.25, True, e16, e8, 0.1, 0.8, per_image=True, pixels=tiles)
        return ind, mask, mode

    ref = highres.compress_tiled(frames, encode_u8, codec)
    got = highres.compress_tiled(frames, encode_u8, codec, chain=True)
    assert len(ref.groups) == 4 and ref.streams() == got.streams()
    # launch counts, through the recorder itself
    shapes = [(c.batch, c.h, c.w) for _, c, _ in got.groups]
    grp = _lib.launch_group(len(got.groups), [b * h * w for b, h, w in shapes], dev)
    with grp as g:
        for k, (_, comp, (ind, masks, mode)) in enumerate(got.groups):
            g.select(k)
            codec.compress(ind, masks, mode)
    assert grp.launches == 1
    grp = _lib.launch_group(len(got.groups), None, dev)
    with grp as g:
        for k, (_, comp, _) in enumerate(got.groups):
            g.select(k)
            codec.decompress(comp, decoder="latency")       # "this call has the GPU to itself"
    assert grp.launches == 1                                # decoder + merge of every group in one launch (decode_merge_kernel)
    grp = _lib.launch_group(len(got.groups), None, dev)
    with grp as g:
        for k, (_, comp, _) in enumerate(got.groups):
            g.select(k)
            codec.decompress(comp)                          # default: the one-launch form only up to half the chip's workgroups
    assert 1 <= grp.launches <= 2 * len(got.groups)
    with cg.decoder_mode("throughput"):                     # the self-synchronising decoder has a grouped form too
        grp = _lib.launch_group(len(got.groups), None, dev)
        with grp as g:
            for k, (_, comp, _) in enumerate(got.groups):
                g.select(k)
                codec.decompress(comp)
    # (the 768x768 group's worst case does not fit the one-workgroup decoder's LDS: it records the split-stream decoder, the other
    # groups the self-synchronising one -- a position with different kernels is launched group by group; the merge is one launch)
    assert 2 <= grp.launches <= len(got.groups) + 1
    torch.cuda.synchronize()


@pytest.mark.gpu
def test_groups_of_mixed_kernels_fall_back_to_single_launches():
    """a 256x256 batch (small-grid kernel variants) grouped with a 768x512 one: positions whose groups recorded different kernels are
    launched one by one -- same results as ungrouped calls"""
    from control_gic_amd import _lib
    from control_gic_amd.quantize import vq_forward_route
    cg, dev, rng, vq, codec = _setup(9)
    xs = [torch.from_numpy(rng.random((5, 3, 256, 256), dtype=np.float32)).to(dev),
          torch.from_numpy(rng.random((2, 3, 512, 768), dtype=np.float32)).to(dev),
          torch.from_numpy(rng.random((1, 3, 64, 48), dtype=np.float32)).to(dev)]
    zs = [torch.from_numpy(rng.standard_normal((x.shape[0], 4, x.shape[2] // 4, x.shape[3] // 4), dtype=np.float32)).to(dev) for x in xs]

    def chain(x, z):
        e8, e16 = cg.entropy_maps(x)
        _, _, ind, mask, _, mode = vq_forward_route(z, vq.embedding.weight, 0.25, True, e16, e8, 0.1, 0.8, per_image=True, pixels=x)
        comp = codec.compress(ind, mask, mode)
        dec = codec.decompress(comp)
        return e8, e16, ind, mask, comp, dec

    ref = [chain(x, z) for x, z in zip(xs, zs)]
    with _lib.launch_group(len(xs), [z.numel() for z in zs], dev) as g:
        got = []
        for k, (x, z) in enumerate(zip(xs, zs)):
            g.select(k)
            got.append(chain(x, z))
    torch.cuda.synchronize()
    for (e8a, e16a, ia, ma, ca, da), (e8b, e16b, ib, mb, cb, db) in zip(ref, got):
        assert torch.equal(e8a, e8b) and torch.equal(e16a, e16b) and torch.equal(ia, ib)
        assert all(torch.equal(p, q) for p, q in zip(ma, mb))
        assert ca.to_host() == cb.to_host()
        assert torch.equal(da[0], db[0]) and torch.equal(da[2], db[2]) and int(db[3].abs().max()) == 0


@pytest.mark.gpu
@pytest.mark.parametrize("H,W", [(1356, 2040), (1001, 1803), (17, 33), (768, 768), (1537, 770)])
@pytest.mark.parametrize("frames", [False, True])
def test_cut_tiles_equals_pad_then_crop(H, W, frames):
    """cgic_cut_tiles (all tiles of all images from the UNPADDED images in one launch) == the reference's F.pad to a multiple of 16
    (centred, zeros; inference_high_resolution.py:145-173,:227-228) followed by the per-tile crops (:236-244), odd pads included"""
    from control_gic_amd import highres
    dev = torch.device("cuda", 0)
    rng = np.random.default_rng(H + 3 * W)
    N = 2
    if frames:
        x = torch.from_numpy(rng.integers(0, 256, (N, H, W, 3), dtype=np.uint8)).to(dev)
        xc = x.permute(0, 3, 1, 2)
    else:
        x = torch.from_numpy(rng.random((N, 3, H, W), dtype=np.float32)).to(dev)
        xc = x
    (left, right, top, bottom), _ = highres.compute_padding(H, W)
    padded = torch.nn.functional.pad(xc, (left, right, top, bottom), mode="constant", value=0)
    tiles = highres.tile_grid(H + top + bottom, W + left + right)
    by_shape = {}
    for i, (_, _, th, tw) in enumerate(tiles):
        by_shape.setdefault((th, tw), []).append(i)
    order = sorted(by_shape.items(), key=lambda kv: -len(kv[1]) * kv[0][0] * kv[0][1])
    batches = highres._cut_all(x, frames, N, H, W, top, left, tiles, order)
    torch.cuda.synchronize()
    for ((th, tw), idxs), batch in zip(order, batches):
        for k, i in enumerate(idxs):
            y, xx, _, _ = tiles[i]
            want = padded[:, :, y:y + th, xx:xx + tw]
            got = batch[:, k].permute(0, 3, 1, 2) if frames else batch[:, k]
            assert torch.equal(got, want), (i, th, tw)


@pytest.mark.gpu
@pytest.mark.parametrize("H,W", [(1000, 1800), (1356, 2040)])
def test_chain_of_a_batch_of_images(H, W):
    """compress_tiled_batch / decompress_tiled_batch with chain=True (N images of one size: every shape group one batch of N x T tiles,
    all groups one launch per kernel) == the ungrouped driver, fp32 images and uint8 frames"""
    from control_gic_amd import highres
    from control_gic_amd.quantize import vq_forward_route
    cg, dev, rng, vq, codec = _setup(H + W)
    N = 3
    frames = torch.from_numpy(rng.integers(0, 256, (N, H, W, 3), dtype=np.uint8)).to(dev)
    x = (frames.permute(0, 3, 1, 2).float() / 255).contiguous()

    def latent(tiles_f32):
        z = torch.nn.functional.avg_pool2d(tiles_f32, 4)
        return (torch.cat([z, z[:, :1] * 2 - 1], dim=1) * 3 - 1.5).contiguous()

    def encode(tiles):
        e8, e16 = cg.entropy_maps(tiles)
        _, _, ind, mask, _, mode = vq_forward_route(latent(tiles), vq.embedding.weight, 0.25, True, e16, e8, 0.1, 0.8, per_image=True, pixels=tiles)
        return ind, mask, mode

    def encode_u8(tiles):
        z = latent(tiles.permute(0, 3, 1, 2).float() / 255)          # (torch work on the INPUT tiles only: allowed inside a chain)
        _, e8, e16 = cg.entropy_maps_u8(tiles, want_x=False)
        _, _, ind, mask, _, mode = vq_forward_route(z, vq.embedding.weight, 0.25, True, e16, e8, 0.1, 0.8, per_image=True, pixels=tiles)
        return ind, mask, mode

    # frames_fp32: ToTensor + pad + crop + maps of the uint8 frames in one pass; `encode` then gets the fp32 tiles (tagged with their maps)
    ref_u8 = highres.compress_tiled_batch(frames, encode_u8, codec)
    got_f = highres.compress_tiled_batch(frames, encode, codec, chain=True, frames_fp32=True)
    for a, b in zip(ref_u8, got_f):
        assert a.streams() == b.streams() and a.bpp() == b.bpp()
    for inp, enc in ((x, encode), (frames, encode_u8)):
        ref = highres.compress_tiled_batch(inp, enc, codec)
        got = highres.compress_tiled_batch(inp, enc, codec, chain=True)
        for a, b in zip(ref, got):
            assert a.streams() == b.streams() and a.bpp() == b.bpp()
        dref = highres.decompress_tiled_batch(ref, codec)
        dgot = highres.decompress_tiled_batch(got, codec, chain=True)
        dcat = highres.decompress_tiled_batch([got[1], got[0]], codec, chain=True)         # not the shared buffers: concatenated first
        for n in range(N):
            _same_decoded(dref[n], dgot[n])
        _same_decoded(dref[1], dcat[0])
        _same_decoded(dref[0], dcat[1])


@pytest.mark.gpu
@pytest.mark.parametrize("H,W", [(1356, 2040), (1001, 1803), (40, 50), (1537, 770)])
@pytest.mark.parametrize("frames", [False, True])
def test_maps_of_tiles_in_one_pass_equal_cut_then_maps(H, W, frames):
    """cgic_entropy_maps_tiles (pad + crop + both entropy maps + flat8 in one pass over the unpadded images; vector loads when the
    window is shifted by a multiple of 4 pixels, element-wise otherwise, zeros inside the pad) == cgic_cut_tiles followed by
    cgic_entropy_maps_f32 / _u8: tiles, maps and flat8 bit for bit; alone and inside a launch group"""
    import control_gic_amd as cg
    from control_gic_amd import highres, _lib
    from control_gic_amd.entropy import entropy_maps_tiles
    dev = torch.device("cuda", 0)
    rng = np.random.default_rng(H * 5 + W)
    N = 2
    if frames:
        x = torch.from_numpy(rng.integers(0, 256, (N, H, W, 3), dtype=np.uint8)).to(dev)
        x[:, : H // 3, : W // 2] = 77                                        # constant patches: flat8 has numbers, not only NaN
    else:
        x = torch.from_numpy(rng.random((N, 3, H, W), dtype=np.float32)).to(dev)
        x[:, :, : H // 3, : W // 2] = 0.25
    (left, right, top, bottom), _ = highres.compute_padding(H, W)
    tiles = highres.tile_grid(H + top + bottom, W + left + right)
    by_shape = {}
    for i, (_, _, th, tw) in enumerate(tiles):
        by_shape.setdefault((th, tw), []).append(i)
    order = sorted(by_shape.items(), key=lambda kv: -len(kv[1]) * kv[0][0] * kv[0][1])
    cut = highres._cut_all(x, frames, N, H, W, top, left, tiles, order)
    ref = []
    for ((th, tw), idxs), batch in zip(order, cut):
        if frames:
            xt, e8, e16 = cg.entropy_maps_u8(batch.view(-1, th, tw, 3))
        else:
            xt = batch.view(-1, 3, th, tw)
            e8, e16 = cg.entropy_maps(xt)
        ref.append((xt, e8, e16, e8._cgic_flat8))

    def fused():
        return [entropy_maps_tiles(x, [(tiles[i][0] - top, tiles[i][1] - left) for i in idxs], th, tw) for (th, tw), idxs in order]

    got_alone = fused()
    with _lib.launch_group(len(order), None, dev) as g:
        got_grouped = []
        for k, ((th, tw), idxs) in enumerate(order):
            g.select(k)
            got_grouped.append(entropy_maps_tiles(x, [(tiles[i][0] - top, tiles[i][1] - left) for i in idxs], th, tw))
    assert g.launches == 1 or len(order) == 1
    torch.cuda.synchronize()
    for got in (got_alone, got_grouped):
        for (xt, e8, e16, f8), (t, g8, g16) in zip(ref, got):
            assert torch.equal(xt, t) and torch.equal(e8, g8) and torch.equal(e16, g16)
            assert torch.equal(torch.nan_to_num(f8, nan=-7.0), torch.nan_to_num(g8._cgic_flat8, nan=-7.0))
            assert cg.entropy_maps(t)[0] is g8                               # the tile batch carries its maps


@pytest.mark.gpu
@pytest.mark.parametrize("B,H,W", [(1, 256, 256), (5, 256, 256), (13, 256, 256), (32, 256, 256), (33, 256, 256), (3, 768, 768), (2, 64, 48), (1, 16, 16)])
def test_decoder_and_merge_in_one_launch_equal_the_throughput_path(B, H, W):
    """small launches in latency mode go out as ONE launch (decode_merge_kernel: the merge bands do their symbol-independent half while
    the split-stream decoder runs and pick the symbols up through a per-image ticket); the throughput decoder keeps two launches.
    Same indices, masks, rows and statuses -- over repeated calls (the ticket resets itself) and all seven routing modes"""
    from control_gic_amd.quantize import vq_forward_route
    cg, dev, rng, vq, codec = _setup(B * 3 + H)
    x = torch.from_numpy(rng.random((B, 3, H, W), dtype=np.float32)).to(dev)
    z = torch.from_numpy(rng.standard_normal((B, 4, H // 4, W // 4), dtype=np.float32)).to(dev)
    e8, e16 = cg.entropy_maps(x)
    for cr, mr in ((0.1, 0.8), (0.0, 0.5), (0.5, 0.0), (0.3, 0.7), (1.0, 0.0), (0.0, 1.0), (0.0, 0.0)):
        _, _, ind, mask, _, mode = vq_forward_route(z, vq.embedding.weight, 0.25, True, e16, e8, cr, mr, per_image=True)
        comp = codec.compress(ind, mask, mode)
        ref = codec.decompress(comp, decoder="throughput")
        for _ in range(3):
            got = codec.decompress(comp, decoder="latency")
            assert torch.equal(ref[0], got[0]) and torch.equal(ref[2], got[2]) and torch.equal(ref[3], got[3])
            assert all(torch.equal(p, q) for p, q in zip(ref[1], got[1]))
        assert int(got[3].abs().max()) == 0
    # a corrupt mask stream is reported by the merge half after the decoder has initialised the status word
    bad = cg.CompressedBatch(comp.data.clone(), comp.nbytes.clone(), comp.mode, comp.h, comp.w)
    _, _, ind, mask, _, mode = vq_forward_route(z, vq.embedding.weight, 0.25, True, e16, e8, 0.1, 0.8, per_image=True)
    comp = codec.compress(ind, mask, mode)
    bad = cg.CompressedBatch(comp.data.clone(), comp.nbytes.clone(), comp.mode, comp.h, comp.w)
    bad.nbytes[0, 4] += 1
    st_l = codec.decompress(bad, decoder="latency")[3]
    st_t = codec.decompress(bad, decoder="throughput")[3]
    assert int(st_l[0]) != 0 and torch.equal(st_l, st_t)


def test_tile_entry_points_check_their_arguments():
    """argument checks of cgic_cut_tiles / cgic_entropy_maps_tiles come before any launch (host logic only)"""
    import ctypes
    from control_gic_amd import _lib
    l = _lib.lib()
    dummy = ctypes.c_void_p(16)                  # never dereferenced: every call below fails its checks first
    bins = _lib.linspace_bins()
    t = (_lib.Tile * 1)(_lib.Tile(16, 0, 0, 0, 16, 18))                       # width not a multiple of 4
    assert l.cgic_cut_tiles(dummy, 0, 1, 16, 16, 1, t, None) == _lib.ERR_INVALID
    t = (_lib.Tile * 1)(_lib.Tile(20, 0, 0, 0, 16, 16))                       # fp32 destination not 16-byte aligned
    assert l.cgic_cut_tiles(dummy, 0, 1, 16, 16, 1, t, None) == _lib.ERR_INVALID
    assert l.cgic_cut_tiles(dummy, 0, 1, 16, 16, 97, t, None) == _lib.ERR_UNSUPPORTED          # more tiles than one launch takes
    assert l.cgic_cut_tiles(dummy, 0, 1, 16, 16, 0, t, None) == _lib.ERR_UNSUPPORTED
    org = (ctypes.c_int * 2)(0, 0)
    args = lambda T, th, tw, x_out=dummy: (dummy, 0, 1, 64, 64, T, org, th, tw, bins, 32, 0.01, x_out, None, None, None, None)
    assert l.cgic_entropy_maps_tiles(*args(49, 16, 16)) == _lib.ERR_UNSUPPORTED                # tiles per image of one shape
    assert l.cgic_entropy_maps_tiles(*args(1, 16, 24)) == _lib.ERR_INVALID                     # tile not a multiple of 16
    assert l.cgic_entropy_maps_tiles(*args(1, 16, 16, None)) == _lib.ERR_INVALID               # the tile batch is required
    assert l.cgic_entropy_maps_tiles(dummy, 0, 1, 64, 64, 1, org, 16, 16, bins, 31, 0.01, dummy, None, None, None, None) == _lib.ERR_UNSUPPORTED
    assert l.cgic_entropy_maps_tiles(dummy, 0, 1, 64, 64, 1, org, 16, 16, bins, 32, 0.5, dummy, None, None, None, None) == _lib.ERR_UNSUPPORTED
    assert l.cgic_entropy_maps_tiles(dummy, 0, 0, 64, 64, 1, org, 16, 16, bins, 32, 0.01, dummy, None, None, None, None) == 0   # empty batch
