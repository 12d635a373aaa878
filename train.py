#!/usr/bin/env python3
"""CMLPL training driver on MI355X -- same command line as the reference's ``train.py``
(flags of train.py:356-379, printed line of train.py:282-289), with the per-step hot path
(train.py:150-278) executed by ``cmlpl_amd.TrainEngine`` (hand-written gfx950 kernels).

Differences from the reference that are deliberate, MI355X-first choices:
  * the labelled / unlabelled splits live in HBM for the whole run and a batch is a list of row INDICES into them
    (the epoch's permutation, also resident): the kernels read the rows where they lie, no batch tensor is gathered
    (the reference copies every batch over PCIe and draws noise on the CPU);
  * ``--graph``: the step is captured once as a hipGraph and replayed, one launch per step, its per-step scalars
    coming from a device-side table filled once per epoch (several GPUs: one graph per stage of the sharded step,
    the collectives eager between them -- cmlpl_amd.distributed.DistStepGraph);
  * noise and dropout come from in-kernel counter-based streams (PCG4D hash + Box-Muller for the augmentation noise,
    Philox4x32-10 for the dropout masks) seeded with the reference's seed 1088;
  * the five logged scalars of every step (loss_hist, train.py:136,274-278) are written by the step into a
    device-side ring and read back once per ``print_per_batches`` steps, not five times a step; the printed line
    is the mean over that window, as in train.py:281-289.
``--synthetic SHAPE`` (B2 | P | B4 | B5) runs without the datasets, which are not shipped.
Multi-GPU: ``python -m torch.distributed.run --nproc-per-node N train.py ...`` shards every batch by
sample over the ranks (cmlpl_amd.distributed); batch sizes must be multiples of N, and a short last batch
is cut to the largest equal shards (see shard_plan)."""
import argparse
import os
import time

import numpy as np
import torch

from cmlpl_amd import HyperParams, NetShape
from hsi_loader import HSIDataSet, SyntheticHSIDataSet, SyntheticScene
from tools.hyper_tools import CalAccuracy, test_whole
from tools.models import BaseNet2

DATASETS = {1: (9, 103), 2: (16, 204), 3: (15, 144), 4: (16, 200)}     # num_classes, num_features (train.py:75-90)
SYNTH = {"B2": (103, 11, 11, 103, 9), "P": (60, 20, 20, 103, 9), "B4": (200, 11, 11, 200, 16),
         "B5": (48, 15, 15, 48, 20)}


class DeviceLoader:
    """shuffle=True DataLoader semantics (one random permutation per epoch, last short batch kept)
    over arrays that already sit in HBM.  A batch is (offset, size) into ``self.perm``, the epoch's permutation in a
    FIXED device buffer (a captured step keeps reading the same buffer): the step takes the rows by index."""

    def __init__(self, arrays, batch_size, generator):
        self.XP, self.X, self.Y = arrays
        self.bs, self.g = batch_size, generator
        self.perm = torch.zeros(len(self.X), dtype=torch.int64, device=self.X.device)

    def __len__(self):
        return (len(self.X) + self.bs - 1) // self.bs

    def __iter__(self):
        # (stream-ordered behind the previous epoch's steps, which still read the buffer)
        self.perm.copy_(torch.randperm(len(self.X), generator=self.g))
        for i in range(0, len(self.X), self.bs):
            yield i, min(self.bs, len(self.X) - i)

    def rows(self, off, size):
        """the gathered batch (engines that do not take indices: the CPU stand-in of the tests)"""
        idx = self.perm[off:off + size]
        return self.XP[idx], self.X[idx], self.Y[idx]


def shard_plan(bt, btu, world):
    """How a GLOBAL batch of bt + btu rows is taken by `world` ranks: (bt_l, btu_l) rows per rank, or None when the
    batch cannot be sharded equally.  Decided from the global sizes only, so every rank decides the same way (a rank
    that skipped a step others ran would leave them waiting in a collective).  Equal shards are required because
    the ranks' loss shares are summed into global means; the rows that do not divide are dropped,
    the pointer/step bookkeeping still advances on every rank alike."""
    bt_l, btu_l = bt // world, btu // world
    if bt_l < 1 or btu_l < 1:
        return None
    return bt_l, btu_l


def main(args, make_engine=None, device=None):
    """``make_engine`` / ``device`` are test hooks (tests/test_train_loop_gloo.py runs this loop as two gloo ranks
    on CPU around a stand-in engine); the product path leaves them None."""
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    if device is None:
        device = torch.device("cuda", int(os.environ.get("LOCAL_RANK", "0")))
        torch.cuda.set_device(device)
    torch.manual_seed(1088)                                         # seed_torch(), train.py:50-58
    if args.synthetic:
        shape = SYNTH[args.synthetic]
        num_classes, num_features = shape[4], shape[3]
        labeled = SyntheticHSIDataSet(shape, args.num_unlabel, 'label', seed=1)
        unlabeled = SyntheticHSIDataSet(shape, args.num_unlabel, 'unlabel', seed=2)
        if not args.no_eval:
            whole = SyntheticScene(shape, 64, 64, seed=3)                  # a 64 x 64 scene cube, every pixel a test pixel
            Y_test, test_array = whole.Y.numpy(), np.arange(len(whole))
    else:
        num_classes, num_features = DATASETS[int(args.dataID)]
        labeled = HSIDataSet(int(args.dataID), 'label', max_iters=args.num_unlabel)
        unlabeled = HSIDataSet(int(args.dataID), 'unlabel', max_iters=args.num_unlabel, num_unlabel=args.num_unlabel)
        if not args.no_eval:
            whole = HSIDataSet(int(args.dataID), 'wholeset')
            test_array = np.load(labeled.root + 'test_array.npy')
            Y_test = (np.load(labeled.root + 'Y.npy') - 1)[test_array]
        shape = (labeled.XP.shape[1], labeled.XP.shape[2], labeled.XP.shape[3], num_features, num_classes)

    hp = HyperParams(lr=args.lr, num_epochs=args.num_epochs, thr=args.thr, alpha=args.alpha,
                     queue_batch=args.queue_batch, temperature=args.temperature, dropout=args.dropout,
                     noise=args.noise)
    bt, btu = args.labeled_batch_size, args.unlabeled_batch_size
    ppb = args.print_per_batches
    if world > 1 and (bt % world or btu % world):
        raise SystemExit(f"--labeled_batch_size {bt} / --unlabeled_batch_size {btu} must be multiples of the "
                         f"number of GPUs ({world}): the batch is sharded equally by sample")
    if make_engine is not None:
        eng = make_engine(NetShape(*shape), bt // world, btu // world, hp, ppb)
    elif world > 1:
        from cmlpl_amd.distributed import DistTrainEngine, init_distributed
        dist = init_distributed("nccl", device)      # checked start-up: one device per local rank, no silent hang
        eng = DistTrainEngine(NetShape(*shape), bt // world, btu // world, hp, device=device, seed=1088, hist_rows=ppb)
    else:
        from cmlpl_amd import TrainEngine
        eng = TrainEngine(NetShape(*shape), bt, btu, hp, device=device, seed=1088, hist_rows=ppb)
    eng.init_params_default(1088)

    gen = torch.Generator().manual_seed(1088)                        # same permutations on every rank
    lab_loader = DeviceLoader(labeled.device_arrays(device), bt, gen)
    unl_loader = DeviceLoader(unlabeled.device_arrays(device), btu, gen)
    num_batches = min(len(lab_loader), len(unl_loader))              # train.py:134
    num_steps = args.num_epochs * num_batches                        # train.py:135
    loss_hist = np.zeros((num_steps, 5))                             # train.py:136
    index_i = -1
    pending = []                      # loss_hist rows of the steps run since the last read-back of the device ring

    def read_back():
        if pending:
            loss_hist[pending] = eng.loss_window(len(pending))
            pending.clear()
    by_index = getattr(eng, "takes_indices", False)
    # (a split smaller than one batch has no full batch to replay: such a run stays eager.  Several GPUs: the sharded
    #  step is replayed stage by stage, its collectives eager in between -- DistStepGraph; offsets are this rank's)
    use_graph = bool(args.graph) and by_index and hasattr(eng, "capture") and len(lab_loader.X) >= bt and len(unl_loader.X) >= btu
    gbt, gbtu = bt // world, btu // world                              # rows of a replayed step on this rank
    graph = None
    t_start = time.time()
    t_warm, steps_warm = t_start, 0
    for epoch in range(args.num_epochs):                             # train.py:146
        batches = list(zip(lab_loader, unl_loader))                  # (offset, size) pairs; draws this epoch's permutations
        if use_graph and graph is not None:
            # the whole epoch's per-step scalars go to the device table at once; full batches are replays
            graph.program([(epoch, bi, lo + rank * gbt, uo + rank * gbtu) for bi, ((lo, ls), (uo, us)) in enumerate(batches)
                           if ls == bt and us == btu])
        for batch_index, ((lo, ls), (uo, us)) in enumerate(batches):
            index_i += 1                                             # train.py:150
            bl, bul, r = ls, us, 0
            if world > 1:                                            # shard by sample; decided on GLOBAL sizes
                plan = shard_plan(ls, us, world)
                if plan is None:      # fewer rows than ranks: every rank skips alike (row stays zero in loss_hist)
                    continue
                bl, bul = plan
                r = rank
            if graph is not None and ls == bt and us == btu:
                graph.launch()
            elif by_index:
                eng.step(lab_loader.XP, lab_loader.X, lab_loader.Y, unl_loader.XP, unl_loader.X, epoch, batch_index,
                         lab_idx=lab_loader.perm[lo + r * bl:lo + (r + 1) * bl],
                         unl_idx=unl_loader.perm[uo + r * bul:uo + (r + 1) * bul])
            else:
                XPl, Xl, Yl = (t[r * bl:(r + 1) * bl].contiguous() for t in lab_loader.rows(lo, ls))
                XPu, Xu, _ = (t[r * bul:(r + 1) * bul].contiguous() for t in unl_loader.rows(uo, us))
                eng.step(XPl, Xl, Yl, XPu, Xu, epoch, batch_index)
            if use_graph and graph is None:
                # the first step ran eagerly (it sets the kernels' attributes); capture now and hand the rest of this
                # epoch's full batches to the graph
                graph = eng.capture(lab_loader.XP, lab_loader.X, lab_loader.Y, unl_loader.XP, unl_loader.X,
                                    lab_loader.perm, unl_loader.perm, gbt, gbtu, capacity=max(num_batches, 1))
                rest = [(epoch, bi, lo2 + rank * gbt, uo2 + rank * gbtu) for bi, ((lo2, ls2), (uo2, us2)) in enumerate(batches)
                        if bi > batch_index and ls2 == bt and us2 == btu]
                if rest:
                    graph.program(rest)
            pending.append(index_i)
            if (batch_index + 1) % ppb == 0 or len(pending) == ppb:
                read_back()           # the five scalars of train.py:274-278 of those steps: one sync, not five per step
            if (batch_index + 1) % ppb == 0:                         # train.py:281-289 (means over the window)
                w = loss_hist[index_i - ppb + 1:index_i + 1]
                if rank == 0:
                    print('Epoch %d/%d:  %d/%d loss_contrast= %.2f total_loss = %.4f cls_loss = %.4f con_loss = %.4f '
                          'acc = %.2f\n' % (epoch + 1, args.num_epochs, batch_index + 1, num_batches,
                                            np.mean(w[:, 0]), np.mean(w[:, 1]), np.mean(w[:, 2]), np.mean(w[:, 3]),
                                            np.mean(w[:, 4]) * 100))
        read_back()                   # rows of the epoch's tail (num_batches % print_per_batches steps)
        if epoch == 0:                # (the read-back has drained the device) what follows runs on warm kernels
            t_warm, steps_warm = time.time(), eng.step_count
    if device.type == "cuda":
        torch.cuda.synchronize()
    if rank == 0:
        steps = eng.step_count
        t_end = time.time()
        print('training: %d steps in %.3f s' % (steps, t_end - t_start))
        if steps > steps_warm:        # the first epoch carries the one-time loading of the kernels
            print('after the first epoch: %d steps in %.3f s = %.4f ms/step' %
                  (steps - steps_warm, t_end - t_warm, (t_end - t_warm) / (steps - steps_warm) * 1e3))
        if args.save_loss_hist:
            np.save(args.save_loss_hist, loss_hist)
    if rank == 0 and not args.no_eval:
        # whole-image inference + accuracy (train.py:291-306).  The scene stays in HBM as its cube and the forward gathers
        # the windows itself (cmlpl_infer_cube): no 19.9 GB patch tensor, no DataLoader (train.py:291-294 streams the
        # materialised patches).  Window shapes the per-sample forward does not take, or a dataset directory without the
        # cube, fall back to the loader.  The source is built ONCE for both networks, and its load time is printed
        # (the reference's "inference time" includes streaming the data).
        from cmlpl_amd.infer import infer_supported
        t_src = time.time()
        source = None
        if infer_supported(NetShape(*shape)):
            source = whole.cube_source(device) if args.synthetic else whole.cube_source(device, dataID=args.dataID)
        if source is None:
            if args.synthetic:      # (cut the scene's windows on the device, then the reference's loader path)
                from cmlpl_amd.patches import extract_patches
                cs = whole.cube_source(device)
                XPw = extract_patches(cs.cube, torch.arange(len(whole), device=device), shape[1]).cpu()
                source = torch.utils.data.DataLoader(torch.utils.data.TensorDataset(XPw, whole.X),
                                                     batch_size=args.val_batch_size, shuffle=False)
            else:
                source = torch.utils.data.DataLoader(whole, batch_size=args.val_batch_size, shuffle=False)
        if device.type == "cuda":
            torch.cuda.synchronize()
        print('evaluation source ready in %.3f s' % (time.time() - t_src))
        for net in range(2):
            model = BaseNet2(num_features=num_features, dropout=args.dropout, num_classes=num_classes,
                             in_channels=shape[0], window=shape[1]).to(device)
            model.load_state_dict(eng.state_dict(net))
            t1 = time.time()
            pred = test_whole(model, source, print_per_batches=10 ** 9)
            OA, Kappa, producerA = CalAccuracy(pred[test_array], Y_test)
            tag = '' if net == 0 else '1'
            print('inference time == %.3f s' % (time.time() - t1))
            print('Result:\n OA%s=%.2f,Kappa=%.2f' % (tag, OA * 100, Kappa * 100))
            print('producerA%s:' % tag, producerA * 100)
            print('AA%s=%.2f' % (tag, np.mean(producerA) * 100))
    if world > 1 and make_engine is None:
        torch.distributed.barrier()
        torch.distributed.destroy_process_group()
    return loss_hist


def build_parser():
    parser = argparse.ArgumentParser()
    parser.add_argument('--dataID', type=int, default=1)
    parser.add_argument('--num_label', type=int, default=5)
    parser.add_argument('--save_path_prefix', type=str, default='./')
    # train
    parser.add_argument('--labeled_batch_size', type=int, default=128)
    parser.add_argument('--unlabeled_batch_size', type=int, default=128)
    parser.add_argument('--val_batch_size', type=int, default=512)
    parser.add_argument('--num_workers', type=int, default=1)
    parser.add_argument('--lr', type=float, default=5e-4)
    parser.add_argument('--num_epochs', type=int, default=20)
    parser.add_argument('--print_per_batches', type=int, default=10)
    parser.add_argument('--num_unlabel', type=int, default=10000)
    parser.add_argument('--thr', type=float, default=1, help='pseudo label threshold')
    parser.add_argument('--alpha', type=float, default=0.95)
    parser.add_argument('--queue-batch', type=float, default=17, help='number of batches stored in memory bank')
    parser.add_argument('--temperature', default=0.3, type=float, help='softmax temperature')
    # network
    parser.add_argument('--teacher_alpha', type=float, default=0.95)
    parser.add_argument('--dropout', type=float, default=0.8)
    parser.add_argument('--noise', type=float, default=0.5)
    parser.add_argument('--m', type=int, default=5, help='number of stochastic augmentations')
    # this build
    parser.add_argument('--synthetic', choices=sorted(SYNTH), default=None,
                        help='run on seeded synthetic patches of this shape (datasets are not shipped)')
    parser.add_argument('--save_loss_hist', default=None, help='write loss_hist [num_steps,5] (train.py:136) as .npy')
    parser.add_argument('--no_eval', action='store_true', help='skip the whole-image inference after training')
    parser.add_argument('--graph', action='store_true',
                        help='capture the training step once as a hipGraph and replay it (several GPUs: the sharded '
                             'step as seven stage graphs, its collectives eager between them)')
    return parser


if __name__ == '__main__':
    main(build_parser().parse_args())
