#!/usr/bin/env python3
"""Child program of tests/test_gpu_zz_dp.py: ONE data-parallel training step of the drop-in PoseNet_GNNskip on `WORLD_SIZE` ranks
sharing one MI355X, launched by `python -m torch.distributed.run` (gloo by default: a 1-GPU box has no second device for RCCL;
CHECKERPOSE_BENCH_BACKEND=nccl runs the same checks over RCCL on a multi-GPU node).  Every rank checks and prints:

  1. replicas that were seeded DIFFERENTLY start the step with rank 0's parameters and BatchNorm buffers (broadcast on the first
     train step, model/_runtime.py:_run_train);
  2. the bucketed, asynchronous all-reduce issued between the backward's hipGraph segments (parallel.py:
     backward_with_bucketed_allreduce_) gives the mean over ranks of the single-replica gradients (to 1e-5 of the largest
     gradient: float atomics in the tiny layers' weight gradients are not run-to-run reproducible) (a second replica with
     dp_allreduce off computes the local gradient; its explicit all-reduce is the reference) -- also on the hipGraph replay;
  3. after the optimizer step all ranks hold identical parameters.
"""
import os
import sys

import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    world, rank = int(os.environ["WORLD_SIZE"]), int(os.environ["RANK"])
    backend = os.environ.get("CHECKERPOSE_BENCH_BACKEND", "gloo")
    ndev = max(torch.cuda.device_count(), 1)
    dev = torch.device("cuda", int(os.environ.get("LOCAL_RANK", "0")) % ndev)
    torch.cuda.set_device(dev)
    if backend == "nccl":
        dist.init_process_group("nccl", device_id=dev)
    else:
        dist.init_process_group(backend)
    from checkerpose_amd.losses.code_loss import MaskedCodeLoss, UnmaskedCodeLoss
    from checkerpose_amd.losses.mask_loss import MaskLoss_interpolate
    from checkerpose_amd.synthetic import build_net, det_image, det_tensor
    B, N = 2, 512
    dtype = os.environ.get("DP_CHILD_DTYPE", "fp32")
    img = det_image(B, seed=100 + rank).to(dev)
    roi_gt = (det_tensor("t_roi", (B, 1, N), seed=rank) > -0.5).float().to(dev)
    x_gt = (det_tensor("t_x", (B, 16, N), seed=rank) > 0).float().to(dev)
    y_gt = (det_tensor("t_y", (B, 16, N), seed=rank) > 0).float().to(dev)
    m_vis = (det_tensor("t_mv", (B, 128, 128), seed=rank) > 0).float().to(dev)
    roi_loss, bit_loss, seg_loss = UnmaskedCodeLoss("BCE"), MaskedCodeLoss("BCE"), MaskLoss_interpolate()

    def loss_of(net):
        roi, xb, yb, seg, _, _ = net(img, None, 3)
        nb = xb.shape[1]
        return roi_loss(roi, roi_gt) + bit_loss(xb, x_gt[:, :nb], roi_gt) + bit_loss(yb, y_gt[:, :nb], roi_gt) + seg_loss(seg[:, 0:1], m_vis)

    def flat(ts):
        return torch.cat([t.detach().reshape(-1).float() for t in ts])

    def same_on_all_ranks(v, what):
        v = v.detach().float().cpu() if backend != "nccl" else v.detach().float()
        got = [torch.empty_like(v) for _ in range(world)]
        dist.all_gather(got, v)
        for r in range(1, world):
            assert torch.equal(got[0], got[r]), "%s differs between rank 0 and rank %d" % (what, r)

    # -- replicas seeded differently: the first train step must start from rank 0's parameters / buffers
    net = build_net(npoint=N, seed=1 + rank).to(dev).train().set_compute_dtype(dtype)
    ref = build_net(npoint=N, seed=1).to(dev).train().set_compute_dtype(dtype)       # what rank 0 holds
    ref.dp_allreduce = False
    opt = torch.optim.SGD(net.parameters(), lr=1e-3)
    for step in range(3):                              # 0: eager replay, 1: hipGraph capture, 2: replay
        opt.zero_grad(set_to_none=True)
        l = loss_of(net)
        if step == 0:
            same_on_all_ranks(flat(net.parameters()), "parameters after the first-step broadcast")
            # (BatchNorm running statistics were broadcast too, but this forward has already updated them with each replica's
            # own batch statistics -- per-replica BatchNorm, as in the reference: no SyncBN there)
        l.backward()
        # the single-replica gradient of the same weights on this rank's shard, reduced explicitly = the reference
        with torch.no_grad():                          # in place: the training program reads the live weights, nothing is rebuilt
            for q, p in zip(ref.parameters(), net.parameters()):
                q.copy_(p)
        for p_ in ref.parameters():
            p_.grad = None
        loss_of(ref).backward()
        live = [(p, q) for p, q in zip(net.parameters(), ref.parameters()) if q.grad is not None]
        assert len(live) > 1000 and all(p.grad is not None for p, _ in live)
        g_ref = flat(q.grad for _, q in live)
        gr = g_ref.cpu() if backend != "nccl" else g_ref
        dist.all_reduce(gr, op=dist.ReduceOp.SUM)
        gr = (gr / world).to(dev)
        g_dp = flat(p.grad for p, _ in live)
        # (not bit for bit: tiny layers add their weight-gradient partials with float atomics, whose order differs run to run)
        err, scale = float((g_dp - gr).abs().max()), float(gr.abs().max())
        assert err <= 1e-5 * scale, "step %d: bucketed async all-reduce != mean of the replicas' gradients (max diff %.3e, scale %.3e)" % (
            step, err, scale)
        assert float((g_dp - g_ref).abs().max()) > 0, "the ranks' shards differ, so must their local gradients"
        opt.step()
        same_on_all_ranks(flat(net.parameters()), "parameters after optimizer step %d" % step)   # identical: every rank applies the SAME reduced buffer
    torch.cuda.synchronize()
    dist.barrier()
    if rank == 0:
        print("DP_STEP_OK world=%d backend=%s dtype=%s segments=%d" % (world, backend, dtype, len(list(net._train_programs.values())[0]["segments"])), flush=True)
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
