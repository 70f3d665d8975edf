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
