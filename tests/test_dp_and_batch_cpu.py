"""CPU: data-parallel host logic over gloo (world_size 2) and the augmentation index map.

The HIP kernels cannot run here, so per-rank gradients come from the CPU oracle (test infrastructure); what is
under test is the product's DP logic (`ssie_amd.dp`): sharding, the single flat all-reduce, the 1/world scale —
"k-rank averaged gradients == 1-rank gradients on the concatenated batch" (SURVEY §8(e)), rel-L2 <= 1e-5.
"""
import os
import socket

import numpy as np
import pytest
import torch
import torch.multiprocessing as mp

from oracle import ssie_oracle as O

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close(); return p


def _worker(rank, world, port, bands, hw, out_dir):
    import sys
    sys.path.insert(0, ROOT)
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    torch.set_num_threads(2)
    import ssie
    ssie.load()
    from ssie_amd import dp, hostlib as H
    r, w, _ = dp.init_from_env("gloo")
    assert (r, w) == (rank, world)
    table, total = H.param_table(bands)
    P = O.closed_form_params(bands)
    x = O.synthetic_patches(4, bands, hw, hw)                     # global batch 4
    mine = dp.shard_range(4, rank, world)
    xs = x[mine.start:mine.stop]
    _, grads, _ = O.loss_and_grads(P, xs, O.JYU_COEFS)
    flat = torch.zeros(total)
    for name, off, shape in table:
        flat[off:off + grads[name].numel()] = grads[name].reshape(-1)
    scale = dp.allreduce_flat_(flat, world)
    flat *= scale
    if rank == 0:
        torch.save(flat, os.path.join(out_dir, "dp_flat.pt"))
    torch.distributed.barrier()
    torch.distributed.destroy_process_group()


def test_dp_two_ranks_equal_single_rank_on_concatenated_batch(tmp_path):
    bands, hw = 5, 16
    port = _free_port()
    mp.spawn(_worker, args=(2, port, bands, hw, str(tmp_path)), nprocs=2, join=True)
    flat = torch.load(os.path.join(tmp_path, "dp_flat.pt"), weights_only=True)
    import ssie
    ssie.load()
    from ssie_amd import hostlib as H
    table, total = H.param_table(bands)
    P = O.closed_form_params(bands)
    x = O.synthetic_patches(4, bands, hw, hw)
    _, grads, _ = O.loss_and_grads({k: v.double() for k, v in P.items()}, x.double(), O.JYU_COEFS)
    _, g32, _ = O.loss_and_grads(P, x, O.JYU_COEFS)
    for name, off, shape in table:
        if name.endswith("k_linear.bias"):
            continue
        got = flat[off:off + int(np.prod(shape))].double()
        ref = grads[name].reshape(-1)
        # NOTE the Fourier / smoothness means are per-batch means of |.|: the rank average of shard gradients equals the
        # gradient of the global mean exactly in exact arithmetic; fp32 sign flips bound the agreement (see test_plan_gpu)
        tol = max(1e-5, 2.0 * (g32[name].reshape(-1).double() - ref).norm().item() / max(ref.norm().item(), 1e-30))
        assert (got - ref).norm().item() <= tol * ref.norm().item() + 1e-12, (name, tol)


def _adam_double(params, grads, m, v, step, lr, grad_scale=1.0, b1=0.9, b2=0.999, eps=1e-8):
    """test double for the HIP Adam launch (hostlib.adam_step) with the same argument contract, in place on the views"""
    g = grads * grad_scale
    m.mul_(b1).add_(g, alpha=1 - b1); v.mul_(b2).addcmul_(g, g, value=1 - b2)
    params.sub_((lr / (1 - b1 ** step)) * m / (v.sqrt() / np.sqrt(1 - b2 ** step) + eps))


def _tail_worker(rank, world, port, bands, hw, out_dir, frozen):
    import sys
    sys.path.insert(0, ROOT)
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    torch.set_num_threads(2)
    import ssie
    ssie.load()
    from ssie_amd import dp, hostlib as H, model as M
    dp.init_from_env("gloo")
    H.adam_step = _adam_double                                     # the only HIP launch in the tail
    net = M.LowLightEnhance(input_channels=bands, lr=1e-3)
    P = O.closed_form_params(bands)
    net.load_state_dict(P)
    net._ensure_device_layout()
    net.set_decomposition_frozen(frozen)
    x = O.synthetic_patches(4, bands, hw, hw)
    mine = dp.shard_range(4, rank, world)
    _, grads, _ = O.loss_and_grads(P, x[mine.start:mine.stop], O.JYU_COEFS)
    for (name, off, shape), p in zip(net._table, net._plist):
        net._gflat[off:off + p.numel()] = grads[name].reshape(-1)          # what the HIP backward leaves in the flat buffer
    net._finish_step(world)                                              # PRODUCT CODE: frozen range -> all-reduce -> scaled Adam
    if rank == 0:
        torch.save({k: v.detach().clone() for k, v in net.state_dict().items()}, os.path.join(out_dir, f"tail_{int(frozen)}.pt"))
    torch.distributed.barrier()
    torch.distributed.destroy_process_group()


@pytest.mark.parametrize("frozen", [False, True])
def test_train_step_tail_two_ranks(tmp_path, frozen):
    """`LowLightEnhance._finish_step` (the part of train_step after the HIP backward) over gloo, world 2: parameters after the
    step == one Adam step on the gradient of the concatenated batch; with a frozen DecompositionNet its parameters do not move
    (model.py:274-279).  The Adam launch is replaced by a torch double with the same argument contract."""
    bands, hw = 5, 16
    port = _free_port()
    mp.spawn(_tail_worker, args=(2, port, bands, hw, str(tmp_path), frozen), nprocs=2, join=True)
    got = torch.load(os.path.join(tmp_path, f"tail_{int(frozen)}.pt"), weights_only=True)
    P = O.closed_form_params(bands)
    x = O.synthetic_patches(4, bands, hw, hw)
    _, g0, _ = O.loss_and_grads(P, x[:2], O.JYU_COEFS)
    _, g1, _ = O.loss_and_grads(P, x[2:], O.JYU_COEFS)
    for k in P:
        if frozen and k.startswith("decomposition_net."):
            assert torch.equal(got[k], P[k]), k
            continue
        g = 0.5 * (g0[k] + g1[k])                                   # rank average == gradient of the global batch mean
        m = 0.1 * g; v = 0.001 * g * g
        ref = P[k] - (1e-3 / 0.1) * m / (v.sqrt() / np.sqrt(0.001) + 1e-8)
        assert torch.allclose(got[k], ref, rtol=0, atol=2e-6), k


def test_shard_range_and_errors():
    import ssie
    ssie.load()
    from ssie_amd import dp
    assert list(dp.shard_range(8, 1, 4)) == [2, 3]
    assert [i for r in range(8) for i in dp.shard_range(256, r, 8)] == list(range(256))
    with pytest.raises(ValueError):
        dp.shard_range(10, 0, 4)
    assert dp.rank_seed(41, 3) == 44


def test_augmentation_index_map_matches_reference(golden_dir):
    """ssie_aug_source_index (the map the device kernel uses) vs reference crops+augmentations (aux.npz fixtures)."""
    import ctypes
    import ssie
    ssie.load()
    from ssie_amd import hostlib as H
    L = H.lib()
    g = np.load(os.path.join(golden_dir, "aux.npz"))
    cube = g["aug_cube"]
    P = 16
    for (x0, y0, mode) in g["aug_crops"]:
        ref = g["aug_patch_%d" % mode]
        out = np.zeros_like(ref)
        si = ctypes.c_int(); sj = ctypes.c_int()
        for i in range(P):
            for j in range(P):
                assert L.ssie_aug_source_index(int(mode), P, i, j, ctypes.byref(si), ctypes.byref(sj)) == 0
                out[i, j] = cube[x0 + si.value, y0 + sj.value]
        assert np.array_equal(out, ref), mode


def _crop_worker(rank, world, port, out_dir):
    import sys
    sys.path.insert(0, ROOT)
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    import ssie
    ssie.load()
    from ssie_amd import dp, harness
    r, w, _ = dp.init_from_env("gloo")
    shapes = [(40, 36, 5), (33, 48, 5), (64, 64, 5)]
    res = {}
    for mode_req, bs in (("shard", 4), ("per_rank", 1), ("auto", 1), ("auto", 4)):
        mode = harness.resolve_dp_mode(mode_req, bs, w)
        np.random.seed(41)                                             # what main.py does on every rank
        rng = np.random.RandomState(dp.rank_seed(41, r)) if mode == "per_rank" else np.random
        steps = [harness.rank_crops(len(shapes), shapes, b, bs, 16, r, w, mode, rng) for b in range(3)]
        res[(mode_req, bs)] = (mode, steps, harness.batches_per_epoch(len(shapes), bs, w, mode))
    gathered = [None] * w
    torch.distributed.all_gather_object(gathered, res)
    if r == 0:
        torch.save(gathered, os.path.join(out_dir, "crops.pt"))
    torch.distributed.barrier()
    torch.distributed.destroy_process_group()


def test_harness_crop_streams_two_ranks(tmp_path):
    """harness data parallelism as SURVEY 8(e) wrote it, over gloo at world 2 (product functions resolve_dp_mode / rank_crops):
    shard mode = the single-process draw (model.py:304-308) cut into contiguous rank slices; per_rank mode = every rank its own
    RNG stream (seed + rank) and its own batch_size samples, cube indices continuing across ranks; auto picks per_rank exactly
    when the reference's batch size does not divide by the rank count."""
    import ssie
    ssie.load()
    from ssie_amd import dp, harness
    port = _free_port()
    mp.spawn(_crop_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    g = torch.load(os.path.join(tmp_path, "crops.pt"), weights_only=False)        # written by this test a moment ago
    shapes = [(40, 36, 5), (33, 48, 5), (64, 64, 5)]
    # shard, batch 4: concatenating the ranks' slices gives the one-process batch, step by step, on the same stream
    np.random.seed(41)
    for b in range(3):
        whole = harness.draw_crops(3, shapes, b, 4, 16)
        assert g[0][("shard", 4)][1][b] + g[1][("shard", 4)][1][b] == whole
    assert g[0][("shard", 4)][0] == "shard" and g[0][("auto", 4)][0] == "shard" and g[0][("auto", 4)][1] == g[0][("shard", 4)][1]
    assert g[0][("shard", 4)][2] == 0 and g[0][("per_rank", 1)][2] == 1          # 3 cubes: 3 // 4 batches; 3 // (1 * 2) steps
    # per_rank, batch 1 (the reference's own batch size on 2 GPUs): own streams, consecutive cube indices
    for r in range(2):
        mode, steps, _ = g[r][("per_rank", 1)]
        assert mode == "per_rank" and g[r][("auto", 1)][0] == "per_rank" and g[r][("auto", 1)][1] == steps
        rng = np.random.RandomState(dp.rank_seed(41, r))
        for b, crops in enumerate(steps):
            assert crops == harness.draw_crops(3, shapes, b * 2 + r, 1, 16, rng)
            assert [c[0] for c in crops] == [(b * 2 + r) % 3]
    assert g[0][("per_rank", 1)][1] != g[1][("per_rank", 1)][1]
    with pytest.raises(ValueError):
        harness.resolve_dp_mode("shard", 1, 2)
    assert harness.resolve_dp_mode("per_rank", 3, 1) == "shard"                   # one process: nothing to choose
