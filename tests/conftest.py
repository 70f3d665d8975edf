import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with `pytest -m gpu` on the GPU box)")


def pytest_collection_modifyitems(config, items):
    import torch
    if torch.cuda.is_available():
        return
    skip = pytest.mark.skip(reason="no HIP device in this container")
    for it in items:
        if "gpu" in it.keywords:
            it.add_marker(skip)


# tests of cgic_decompress_streams run once per prefix decoder (cgic_set_decode_mode): the split-stream kernels ("latency")
# and the self-synchronising per-image kernel ("throughput") must both match the oracle / the reference's files
BOTH_DECODERS = {
    "test_compress_config1_bit_identical_bins", "test_compress_batch_vs_oracle_and_roundtrip",
    "test_decompress_flags_corrupt_streams", "test_fused_post_quant_conv_is_a_second_gather",
    "test_codec_random_sweep_time_boxed", "test_split_decode_tiny_and_huge_streams_round_trip",
    "test_tiled_compress_matches_reference_files", "test_decoders_agree_on_adversarial_streams", "test_small_tables_round_trip",
}


def pytest_generate_tests(metafunc):
    if metafunc.function.__name__ in BOTH_DECODERS:
        metafunc.parametrize("decoder_under_test", ["latency", "throughput"], indirect=True)


@pytest.fixture(autouse=True)
def decoder_under_test(request):
    mode = getattr(request, "param", None)
    if mode is None:
        yield None
        return
    import control_gic_amd as cg
    with cg.decoder_mode(mode):
        yield mode


@pytest.fixture(autouse=True)
def ticket_slots_come_back_zeroed(request):
    """after every GPU test: the library's ticket memory (zero-on-entry slots that every kernel hands back zeroed) is all zero once
    the device is idle -- a slot left dirty breaks whichever launch gets it next, possibly many tests later (wrong bytes, or
    workgroups that wait forever): this check names the test that left it"""
    yield
    if "gpu" not in request.keywords:
        return
    import torch
    if not torch.cuda.is_available():
        return
    import control_gic_amd as cg
    dirty = cg._lib.lib().cgic_ticket_pool_dirty_words()
    if dirty:
        import ctypes
        buf = (ctypes.c_uint * (4 * 48))()
        n = cg._lib.lib().cgic_ticket_pool_dirty_dump(buf, 48)
        where = [tuple(buf[4 * i + j] for j in range(4)) for i in range(max(n, 0))]
        assert dirty == 0, f"{dirty} words of the ticket pools are not zero after this test: (where, slot, word, value) {where}"


def load_golden(name):
    return dict(np.load(os.path.join(GOLDEN, name + ".npz")))


@pytest.fixture(scope="session")
def golden():
    cache = {}

    def get(name):
        if name not in cache:
            cache[name] = load_golden(name)
        return cache[name]
    return get


@pytest.fixture(scope="session")
def orc():
    """the CPU oracle (test infrastructure; never imported by the product package)"""
    from oracle import cgic_oracle
    cgic_oracle.build()
    return cgic_oracle


def unpack_mask(bits, shape):
    n = int(np.prod(shape))
    return np.unpackbits(bits)[:n].astype(np.int32).reshape(shape)
