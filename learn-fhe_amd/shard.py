"""Data-parallel sharding of independent ring operations across the GPUs of one node.

Batches of independent polynomials / ciphertexts split contiguously across ranks; twiddle tables and
keys are replicated.  No collective is needed on the data path; the only
collective is the optional final gather of results."""
from __future__ import annotations


def shard_range(total: int, rank: int, world: int):
    """Contiguous [lo, hi) slice of `total` independent units owned by `rank`; sizes differ by at most one
    and every unit is owned exactly once (ragged totals allowed, empty shards allowed)."""
    if world <= 0 or not (0 <= rank < world) or total < 0:
        raise ValueError("bad shard request")
    base, rem = divmod(total, world)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def gather_results(local, group=None):
    """Final gather of per-rank results (torch tensors of equal shape) onto every rank: one all_gather
    (RCCL over xGMI on GPUs, gloo on CPU).  Not part of the timed hot path."""
    import torch
    import torch.distributed as dist
    world = dist.get_world_size(group)
    out = [torch.empty_like(local) for _ in range(world)]
    dist.all_gather(out, local, group=group)
    return torch.cat(out, dim=0)


# ---- cfg4: CKKS key switch with the RNS limbs sharded across ranks -------------------------------
#
# `Ckks::key_switch` (scheme/ckks/src/ckks.rs:284-293) is per-limb independent except for the two base conversions.  With
# L = K = world, rank r owns q-limb r and p-limb r:
#   1. ct.a (all L q-limbs, 8 N L bytes) is replicated at staging; rank r extends it to ITS p-limb locally (rns.rs:331-345);
#   2. for its two limbs: forward transform, product with its key limbs (evaluation domain), inverse transform -- local;
#   3. ONE all-gather of the two p-limb products (2 x 8 N bytes per rank): every rank then holds all K p-limbs it needs to
#      base-convert P -> its own q-limb in `rescale_k` (rns.rs:103-118);
#   4. rescale its q-limb (+ ct.b limb), local; the final gather of the 2 x L output limbs is the consumer's choice.
# The arithmetic is injected (`ops`): the GPU backend below drives the C ABI; tests inject the oracle to check the
# orchestration on CPU ranks under gloo.


class GpuLimbOps:
    """Per-rank device arithmetic of the limb-sharded key switch: contexts for q_r, p_r, (qs -> p_r) and (q_r <- ps)."""

    def __init__(self, fhe, qs, ps, rank, device=0):
        self.fhe, self.qs, self.ps, self.rank = fhe, list(qs), list(ps), rank
        self.ctx_q, self.ctx_p = fhe.NttContext(qs[rank], device=device), fhe.NttContext(ps[rank], device=device)
        self.ext = fhe.RnsContext(self.qs, [ps[rank]], device=device)      # all q-limbs -> my p-limb
        self.resc = fhe.RnsContext([qs[rank]], self.ps, device=device)     # my q-limb <- all p-limbs

    def extend_to_my_p(self, a_q_all, n):          # [L][n] -> [n]
        return self.ext.extend_bases(a_q_all.reshape(1, len(self.qs), n), n).reshape(n)

    def limb_product(self, which, x, key_eval, n):  # iNTT(NTT(x) (.) key_eval) over my q- or p-modulus; x is consumed
        ctx = self.ctx_q if which == "q" else self.ctx_p
        ctx.ntt_(x, n)
        ctx.pointwise_mul_(x, key_eval)
        return ctx.intt_(x, n)

    def key_to_eval(self, which, k, n):
        return (self.ctx_q if which == "q" else self.ctx_p).ntt_(k, n)

    def rescale_my_q(self, x_q, x_p_all, n):        # [n], [K][n] -> [n]
        import torch
        stacked = torch.cat([x_q.reshape(1, n), x_p_all.reshape(len(self.ps), n)], dim=0).reshape(1, 1 + len(self.ps), n).contiguous()
        return self.resc.rescale_k(stacked, n).reshape(n)

    def add_my_q(self, x, y):
        return self.fhe.rq_add(self.qs[self.rank], x, y)


def dist_all_gather(x, group=None):
    """Stack one equal-shape tensor per rank -> [world][...]: RCCL on device tensors under the nccl backend (direct peer
    writes over xGMI for these 2 x 8 N-byte messages), through host memory under gloo (CPU ranks, or rehearsals where the
    ranks share one GPU)."""
    import torch
    import torch.distributed as dist
    world = dist.get_world_size(group)
    if dist.get_backend(group) == "nccl":
        out = [torch.empty_like(x) for _ in range(world)]
        dist.all_gather(out, x.contiguous(), group=group)
        return torch.stack(out, dim=0)
    out = [torch.empty_like(x, device="cpu") for _ in range(world)]
    dist.all_gather(out, x.cpu().contiguous(), group=group)
    return torch.stack(out, dim=0).to(x.device)


def ckks_key_switch_limb_sharded(ops, rank, world, n, ct_b_limb, ct_a_all, ksk_b_q, ksk_a_q, ksk_b_p, ksk_a_p, all_gather):
    """One rank's part of the limb-sharded key switch.  ct_b_limb [n]: ct.b's q-limb `rank`; ct_a_all [L][n]: ct.a, replicated;
    ksk_{b,a}_{q,p} [n]: this rank's key limbs already in the evaluation domain (`ops.key_to_eval`); all_gather(x) -> [world][..]
    stacks one tensor per rank (RCCL / gloo).  Returns (b', a') for q-limb `rank`."""
    a_q = ct_a_all[rank].clone()
    a_p = ops.extend_to_my_p(ct_a_all, n)
    pb_q = ops.limb_product("q", a_q.clone(), ksk_b_q, n)
    pa_q = ops.limb_product("q", a_q, ksk_a_q, n)
    pb_p = ops.limb_product("p", a_p.clone(), ksk_b_p, n)
    pa_p = ops.limb_product("p", a_p, ksk_a_p, n)
    import torch
    gathered = all_gather(torch.stack([pb_p, pa_p], dim=0))  # [world][2][n]: the only collective on the path
    b = ops.rescale_my_q(pb_q, gathered[:, 0], n)
    a = ops.rescale_my_q(pa_q, gathered[:, 1], n)
    return ops.add_my_q(b, ct_b_limb), a


# ---- the same partition on the library's own sharded entry points: a BATCH of ciphertexts per call, the fused kernels -----------
#
# fhe_ckks_shard_products / fhe_ckks_shard_finish (include/fhe_ring.h): every rank holds the context and the prepared key and owns
# a contiguous slice of the q-limbs and of the p-limbs; per batch ONE all-gather of the p-limb products ([2][batch][np][n] per rank)
# and nothing else crosses ranks.  Under nccl the gather is RCCL on device buffers with no host synchronisation around it.


def limb_slices(big_l, big_k, rank, world):
    """(q_lo, q_hi, p_lo, p_hi) of `rank`: contiguous, every limb owned once; K must split evenly (the gather's contributions are equal)."""
    if big_k % world:
        raise ValueError("K = %d p-limbs do not split evenly over %d ranks" % (big_k, world))
    q_lo, q_hi = shard_range(big_l, rank, world)
    np_ = big_k // world
    return q_lo, q_hi, rank * np_, (rank + 1) * np_


def all_gather_into(x, group=None):
    """[...] per rank -> [world][...] on every rank.  nccl: one RCCL all-gather on device memory, asynchronous w.r.t. the host;
    gloo (CPU ranks, or rehearsals where the ranks share one GPU): through host memory."""
    import torch
    import torch.distributed as dist
    world = dist.get_world_size(group)
    if dist.get_backend(group) == "nccl":
        out = torch.empty((world,) + tuple(x.shape), dtype=x.dtype, device=x.device)
        dist.all_gather_into_tensor(out, x.contiguous(), group=group)
        return out
    out = [torch.empty_like(x, device="cpu") for _ in range(world)]
    dist.all_gather(out, x.cpu().contiguous(), group=group)
    return torch.stack(out, dim=0).to(x.device)


def ckks_key_switch_sharded_batch(shard, ct_b_mine, ct_a_all, all_gather=all_gather_into):
    """One rank's part for a batch: shard = fhe.CkksShard of this rank; ct_b_mine [batch][nq][n] (ct.b's owned limbs), ct_a_all
    [batch][L][n] (ct.a, replicated) -> (b', a') [batch][nq][n], the owned limbs of the switched ciphertexts."""
    prod_q, prod_p = shard.products(ct_a_all)
    gathered = all_gather(prod_p)  # [world][2][batch][np][n]: the only exchange on the path
    return shard.finish(prod_q, gathered, ct_b_mine)
