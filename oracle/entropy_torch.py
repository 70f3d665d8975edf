"""torch-CPU restatement of the reference's entropy maps (CGIC/models/model.py:433-483), in the reference's own arithmetic.

TEST INFRASTRUCTURE ONLY (imported by tests/, bench.py's mask-flip report): never by the product package.

Why a second restatement next to cgic_oracle.c: the router thresholds are k-th smallest entropy values, and on tie-heavy
content (8-bit, flat, blocky images) WHICH patches fall under a threshold depends on the last bits of the entropies --
on torch's vectorised exp / log and on the order in which torch's CPU `mean` adds the 64 or 256 kernel values of a
patch.  The plain-C oracle uses libm and a sequential sum: it agrees with the reference to ~1e-6, not to the bit.  This
module executes the SAME torch CPU operators on tensors of the SAME shapes and strides as the reference (gray ->
unfold p x p -> [patches, p*p, 1] - bins[32] -> /sigma, square, * -0.5, exp -> mean over pixels -> normalise ->
-sum p log p), so on any host with the same torch build it is the reference's result bit for bit.
Parity status: pinned -- tests/golden/make_golden_ties.py runs the real `Entropy` class of /root/reference next to it
on every content family and requires torch.equal (1 and 8 threads) before writing tests/golden/ties.npz.
"""
import torch
import torch.nn.functional as F


def entropy_map(x, p):
    """x [B,3,H,W] fp32 CPU tensor -> [B,H/p,W/p] fp32 (model.py:463-483 + :441-461)"""
    if x.device.type != "cpu" or x.dtype != torch.float32:
        raise TypeError("entropy_torch: CPU fp32 tensor expected (this is the CPU reference arithmetic)")
    B, _, H, W = x.shape
    hn, wn = H // p, W // p
    gray = 0.2989 * x[:, 0:1, :, :] + 0.5870 * x[:, 1:2, :, :] + 0.1140 * x[:, 2:, :, :]           # :471
    cols = F.unfold(gray, kernel_size=(p, p), stride=p)                                              # [B, p*p, P]  :474
    vals = torch.reshape(cols.transpose(1, 2).unsqueeze(2), (B * hn * wn, p * p))                    # :476-478
    bins = torch.linspace(-1, 1, 32)                                                                 # :480
    sigma = torch.tensor(0.01)
    eps = 1e-40                                                                                      # :451
    res = vals.unsqueeze(2) - bins.unsqueeze(0).unsqueeze(0)                                         # :452-453
    kv = torch.exp(-0.5 * (res / sigma).pow(2))                                                      # :454
    pdf = torch.mean(kv, dim=1)                                                                      # :456
    norm = torch.sum(pdf, dim=1).unsqueeze(1) + eps                                                  # :457
    pdf = pdf / norm + eps                                                                           # :458
    ent = -torch.sum(pdf * torch.log(pdf), dim=1)                                                    # :459
    return ent.reshape(B, hn, wn)                                                                    # :460-461


def entropy_maps(x):
    """(e8, e16) like control_gic_amd.entropy_maps"""
    return entropy_map(x, 8), entropy_map(x, 16)
