"""Data-parallel TRAINING STEP of the model: two ranks, each with its own shard of the batch, run
TinyYolo forward + loss + BPTT backward through ``FlatTrainer.step()`` (flat-gradient all-reduce, 1/world inside the
fused Adamax) and must reproduce the 2-rank CPU restatement of the reference's DDP step:

* without SyncBatchNorm: two oracle replicas, one shard each, per-rank loss means, gradients averaged
  (``config/config.yaml:34-37``);
* with SyncBatchNorm (``config/config.yaml:76``): batch statistics of the GLOBAL batch per timestep, per-rank loss
  means, gradients averaged - restated as one oracle pass over the concatenated batch whose loss is the mean of
  the two shard losses.

Both ranks share the one GPU of the test box and exchange through gloo (fresh child processes; nothing is re-exec'd
after the GPU was initialised); the production transport is RCCL, which needs one GPU per rank.
"""
import datetime
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from tests.util import rel_err, synthetic_events, synthetic_labels

pytestmark = pytest.mark.gpu
T, B_RANK, H, W = 4, 2, 32, 48


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _shard(rank):
    return (synthetic_events(T, B_RANK, H, W, p=0.08, seed=10 + rank), synthetic_labels(B_RANK, seed=20 + rank))


def _worker(rank, world, port, out_dir, sync_bn, grad_slots=True):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world, timeout=datetime.timedelta(seconds=180))
    try:
        import snn_for_object_detection_amd as S
        from snn_for_object_detection_amd.trainer import FlatTrainer, broadcast_parameters, convert_sync_batchnorm
        torch.manual_seed(100 + rank)            # ranks start from different weights; the broadcast aligns them
        model = S.TinyYolo(num_classes=2, time_window=0).cuda().train()
        tr = FlatTrainer(model, lr=1e-3, use_grad_slots=grad_slots)
        assert (tr._early_lo is not None) == grad_slots   # no overlapped exchange without slots on the neck / head
        broadcast_parameters(tr)
        if sync_bn:
            convert_sync_batchnorm(model)
        start = {k: v.detach().cpu().clone() for k, v in model.state_dict().items()}
        X, labels = _shard(rank)
        tr.zero_grad()
        loss = model.training_step((X.cuda(), labels.cuda()))
        loss.backward()
        # between backward() and step(): synchronize() joins the weight-gradient stream AND the early exchange, so the
        # neck / head part is already summed over the ranks while the backbone part is still this rank's own
        tr.synchronize()
        if not grad_slots:
            tr._collect_autograd_grads()
        local = {n: g.detach().cpu().clone() for n, g in tr.grads_by_name(model).items()}
        tr.step()
        tr.sync_buffers()                        # rank 0's BatchNorm buffers everywhere (DDP's broadcast_buffers)
        torch.cuda.synchronize()
        avg = {n: (g / world).detach().cpu().clone() for n, g in tr.grads_by_name(model).items()}
        torch.save({"start": start, "loss": loss.item(), "local": local, "avg": avg,
                    "after": {k: v.detach().cpu().clone() for k, v in model.state_dict().items()},
                    "backend": dist.get_backend(), "world": dist.get_world_size(), "overlapped": tr._early_lo is not None},
                   os.path.join(out_dir, f"rank{rank}.pt"))
    finally:
        dist.destroy_process_group()


@pytest.mark.timeout(600)
@pytest.mark.parametrize("sync_bn,grad_slots", [(False, True), (True, True), (False, False)],
                         ids=["rank-local-bn", "sync-bn", "autograd-grads-no-overlap"])
def test_two_rank_training_step_matches_cpu_ddp_restatement(tmp_path, hip_lib, sync_bn, grad_slots):
    if not torch.cuda.is_available():
        pytest.skip("no HIP device")
    import snn_for_object_detection_amd as S
    from oracle.net import SODaRef
    world, port = 2, _free_port()
    mp.spawn(_worker, args=(world, port, str(tmp_path), sync_bn, grad_slots), nprocs=world, join=True)
    r = [torch.load(tmp_path / f"rank{k}.pt") for k in range(world)]
    assert r[0]["world"] == 2
    for k in r[0]["start"]:                                   # broadcast from rank 0
        assert torch.equal(r[0]["start"][k], r[1]["start"][k]), k

    desc = S.TinyYolo(num_classes=2, time_window=0)
    shards = [_shard(k) for k in range(world)]

    def oracle_replica():
        o = SODaRef(desc, 2, time_window=0)
        o.load_state_dict(r[0]["start"])
        return o.train()

    if not sync_bn:
        replicas, losses = [oracle_replica() for _ in range(world)], []
        for o, (X, labels) in zip(replicas, shards):
            loss = o.training_step((X, labels))
            loss.backward()
            losses.append(loss.item())
        want = {n: sum(dict(o.named_parameters())[n].grad for o in replicas) / world
                for n, p0 in replicas[0].named_parameters() if p0.requires_grad}
        bn_ref = replicas[0]          # rank 0's buffers are what every rank holds after the step (DDP broadcast_buffers)
    else:
        o = oracle_replica()
        X = torch.cat([s[0] for s in shards], dim=1)
        preds = o(X)                                          # BatchNorm over the global batch, per timestep
        losses = []
        for k, (_, labels) in enumerate(shards):
            sl = slice(k * B_RANK, (k + 1) * B_RANK)
            losses.append(o._loss((preds[0], preds[1][sl], preds[2][sl]), labels))
        (sum(losses) / world).backward()
        losses = [l.item() for l in losses]
        want = {n: p.grad for n, p in o.named_parameters() if p.requires_grad}
        bn_ref = o
    for k in range(world):
        assert abs(r[k]["loss"] - losses[k]) <= 1e-4 * abs(losses[k]), (k, r[k]["loss"], losses[k])
    worst = 0.0
    for n, g in want.items():
        if g is not None and g.norm() > 1e-8:
            assert torch.equal(r[0]["avg"][n], r[1]["avg"][n]), n            # one all-reduce: identical on both ranks
            worst = max(worst, rel_err(r[0]["avg"][n], g))
    assert worst < 1e-3, worst
    # what a caller reads between backward() and step() (gradient-norm logging, clipping): the part that went out from
    # the backward hook is the rank SUM already, the rest is rank-local and sums to the exchanged gradient
    for n in want:
        early = r[0]["overlapped"] and n.split(".")[0] in ("neck_net", "head_net")
        for k in range(world):
            if early:
                assert torch.equal(r[k]["local"][n] / world, r[k]["avg"][n]), n
        if not early:
            assert torch.equal((r[0]["local"][n] + r[1]["local"][n]) / world, r[0]["avg"][n]), n
    # the optimiser step: torch.optim.Adamax on the oracle's averaged gradient, from the same start
    params = {n: torch.nn.Parameter(r[0]["start"][n].clone()) for n in want}
    for n, p in params.items():
        p.grad = want[n].clone()
    torch.optim.Adamax(list(params.values()), lr=1e-3).step()
    for n, p in params.items():
        assert torch.equal(r[0]["after"][n], r[1]["after"][n]), n
        assert rel_err(r[0]["after"][n], p.detach()) < 1e-5, n
    # BatchNorm buffers: identical on both ranks after the step, equal to the restatement's
    ref_sd = bn_ref.state_dict()
    for k, v in r[0]["after"].items():
        if "running_" in k or "num_batches_tracked" in k:
            assert torch.equal(v, r[1]["after"][k]), k
            if v.is_floating_point():
                assert rel_err(v, ref_sd[k]) < 1e-5, k
            else:
                assert int(v) == int(ref_sd[k]) == T, k


def _uneven_worker(rank, world, port, out_dir):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world, timeout=datetime.timedelta(seconds=120))
    try:
        from snn_for_object_detection_amd import BlockGen, Conv, LIF, Norm
        from snn_for_object_detection_amd.trainer import FlatTrainer, broadcast_parameters
        torch.manual_seed(3)
        blk = BlockGen(2, [Conv(8, 3), Norm(), LIF(), Conv(8, 1)]).cuda().train()
        extra = BlockGen(8, [Conv(4, 1)]).cuda()
        model = torch.nn.ModuleList([blk, extra])
        tr = FlatTrainer(model, lr=1e-2)
        broadcast_parameters(tr)
        x = synthetic_events(3, 2, 12, 16, p=0.3, seed=rank).cuda()
        w_extra = extra.net[0][0].weight
        start = w_extra.detach().clone()
        for it in range(2):
            tr.zero_grad()
            if rank == 0 and it == 0:            # ONLY rank 0 produces a gradient for `extra`, and only in step 0
                w_extra.grad = torch.ones_like(w_extra)
            out, _ = blk(x)
            out.square().mean().backward()
            tr.step()                             # must not hang: every rank enters the same collectives
        torch.cuda.synchronize()
        torch.save({"extra": w_extra.detach().cpu(), "start": start.cpu(), "flat": tr.flat_param.detach().cpu(),
                    "steps": list(tr.param_steps)}, os.path.join(out_dir, f"uneven{rank}.pt"))
    finally:
        dist.destroy_process_group()


@pytest.mark.timeout(300)
def test_a_parameter_written_on_one_rank_only_does_not_hang_the_step(tmp_path, hip_lib):
    """Rank 1 has no gradient for a parameter rank 0 wrote: the written-flag exchange used to sit behind a rank-local
    condition (rank 0 skipped the collective rank 1 waited in).  Now both ranks finish, the parameter moves by the
    AVERAGED gradient on both (written on ANY rank counts) in step 0 and stays put in step 1."""
    if not torch.cuda.is_available():
        pytest.skip("no HIP device")
    world, port = 2, _free_port()
    mp.spawn(_uneven_worker, args=(world, port, str(tmp_path)), nprocs=world, join=True)
    r0, r1 = (torch.load(tmp_path / f"uneven{k}.pt") for k in range(world))
    assert torch.equal(r0["flat"], r1["flat"])                       # replicas stayed together
    assert not torch.equal(r0["extra"], r0["start"])                 # the one-rank gradient arrived everywhere
    assert r0["steps"] == r1["steps"] and r0["steps"][-1] == 1 and r0["steps"][0] == 2
    # Adamax, first step, gradient 1/2 (ones averaged over two ranks): the update is lr * sign = 1e-2 exactly-ish
    assert torch.allclose(r0["start"] - r0["extra"], torch.full_like(r0["start"], 1e-2), rtol=1e-4, atol=0)


def _rccl_worker(rank, world, port, out_dir, sync_bn):
    """One rank on RCCL ("nccl"), every collective of the N > 1 step issued (``exchange_single_rank``)."""
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    torch.cuda.set_device(0)
    dist.init_process_group("nccl", rank=rank, world_size=world, timeout=datetime.timedelta(seconds=180),
                            device_id=torch.device("cuda", 0))
    try:
        import snn_for_object_detection_amd as S
        from snn_for_object_detection_amd.trainer import FlatTrainer, broadcast_parameters, convert_sync_batchnorm
        torch.manual_seed(100)
        model = S.TinyYolo(num_classes=2, time_window=0).cuda().train()
        tr = FlatTrainer(model, lr=1e-3, exchange_single_rank=True)
        assert tr.exchange and tr._early_lo is not None      # the overlapped exchange is armed
        broadcast_parameters(tr)
        if sync_bn:
            convert_sync_batchnorm(model)
        X, labels = _shard(0)
        X, labels = X.cuda(), labels.cuda()
        losses = []
        for _ in range(3):
            tr.zero_grad()
            loss = model.training_step((X, labels))
            loss.backward()
            assert tr._early_work is not None                # the backward hook started the early all-reduce
            tr.step()
            losses.append(loss.item())
        tr.sync_buffers()
        torch.cuda.synchronize()
        torch.save({"losses": losses, "after": {k: v.detach().cpu().clone() for k, v in model.state_dict().items()},
                    "backend": dist.get_backend()}, os.path.join(out_dir, "rccl.pt"))
    finally:
        dist.destroy_process_group()


@pytest.mark.timeout(600)
@pytest.mark.parametrize("sync_bn", [False, True], ids=["rank-local-bn", "sync-bn"])
def test_one_rank_on_rccl_with_every_collective_issued_equals_the_plain_step(tmp_path, hip_lib, sync_bn):
    """The production transport (RCCL) has only gloo rehearsals above.  A one-GPU box cannot hold two RCCL ranks, but it
    can run the N > 1 step's collectives in a ONE-rank RCCL group: broadcast of the weights, the early all-reduce from the
    backward hook on the communication stream, the written-flag MAX all-reduce, the head all-reduce, the SyncBatchNorm
    exchanges, the buffer broadcast.  Sums over one rank change nothing, so three steps must equal the plain
    single-process run bit for bit - what this pins is the stream / event / work-handle plumbing under ProcessGroupNCCL."""
    if not torch.cuda.is_available():
        pytest.skip("no HIP device")
    import snn_for_object_detection_amd as S
    from snn_for_object_detection_amd.trainer import FlatTrainer
    mp.spawn(_rccl_worker, args=(1, _free_port(), str(tmp_path), sync_bn), nprocs=1, join=True)
    got = torch.load(os.path.join(tmp_path, "rccl.pt"))
    assert got["backend"] == "nccl"
    torch.manual_seed(100)
    model = S.TinyYolo(num_classes=2, time_window=0).cuda().train()
    tr = FlatTrainer(model, lr=1e-3)
    X, labels = _shard(0)
    X, labels = X.cuda(), labels.cuda()
    losses = []
    for _ in range(3):
        tr.zero_grad()
        loss = model.training_step((X, labels))
        loss.backward()
        tr.step()
        losses.append(loss.item())
    torch.cuda.synchronize()
    assert got["losses"] == losses
    for k, v in model.state_dict().items():
        assert torch.equal(got["after"][k], v.detach().cpu()), k
