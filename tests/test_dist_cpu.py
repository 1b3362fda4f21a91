"""Batch sharding + detection gather with two gloo ranks on the CPU (the N>1 path of bench.py / yolo_amd.dist)."""
import os
import socket

import numpy as np
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

import util
from yolo_amd import dist as ydist
from yolo_amd.utils import utils as yu


def _free_port():
    with socket.socket() as s:
        s.bind(('127.0.0.1', 0))
        return s.getsockname()[1]


def _pad(out, max_det=300):
    det = torch.zeros(len(out), max_det, 6)
    cnt = torch.zeros(len(out), dtype=torch.int32)
    for b, o in enumerate(out):
        if o is not None:
            det[b, :o.shape[0]] = o
            cnt[b] = o.shape[0]
    return det, cnt


def _worker(rank, world, port, q):
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank), MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port))
    r, w, _ = ydist.init_from_env(backend='gloo')
    z = util.golden('nms_multilabel')
    pred = torch.from_numpy(np.concatenate([z['pred'], z['pred'][::-1]], 0))  # 4 images
    lo, hi = ydist.shard_bounds(pred.shape[0], r, w)
    det, cnt = _pad(yu.non_max_suppression(pred[lo:hi].clone(), float(z['conf']), float(z['iou'])))
    det_all, cnt_all = ydist.gather_detections(det, cnt)
    if r == 0:
        q.put((det_all.numpy(), cnt_all.numpy()))
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_gather_equals_single_process():
    ctx = mp.get_context('spawn')
    q = ctx.SimpleQueue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    det_all, cnt_all = q.get()
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    z = util.golden('nms_multilabel')
    want_det = np.concatenate([z['det'], z['det'][::-1]], 0)
    want_cnt = np.concatenate([z['count'], z['count'][::-1]], 0)
    assert np.array_equal(cnt_all, want_cnt) and np.array_equal(det_all, want_det)
    out = ydist.to_list(torch.from_numpy(det_all), torch.from_numpy(cnt_all))
    assert len(out) == 4 and out[0].shape == (int(want_cnt[0]), 6)


def test_shard_bounds_and_packing():
    assert [ydist.shard_bounds(10, r, 4) for r in range(4)] == [(0, 3), (3, 6), (6, 8), (8, 10)]
    assert [ydist.shard_bounds(256, r, 8) for r in range(8)] == [(32 * r, 32 * r + 32) for r in range(8)]
    det = torch.rand(3, 300, 6)
    cnt = torch.tensor([0, 300, 17], dtype=torch.int32)
    d2, c2 = ydist.unpack_detections(ydist.pack_detections(det, cnt), 300)
    assert torch.equal(d2, det) and torch.equal(c2, cnt)


_RANK_SCRIPT = '''
import os, sys, torch, torch.distributed as dist
sys.path.insert(0, %r)
from yolo_amd import dist as ydist
rank, world, local = ydist.init_from_env(backend='gloo')
t = torch.tensor([float(rank + 1)])
dist.all_reduce(t)
open(os.path.join(%r, 'rank%%d.txt' %% rank), 'w').write('%%d %%d %%g' %% (world, local, t.item()))
print('hello from rank %%d' %% rank)
dist.barrier()
dist.destroy_process_group()
'''


def test_launch_ranks_starts_one_process_per_rank(tmp_path):
    """The launcher behind a plain `python bench.py --gpus N`: rank processes get the torch.distributed.run environment,
    rendezvous on 127.0.0.1 and all exit 0."""
    import sys
    script = tmp_path / 'rank.py'
    script.write_text(_RANK_SCRIPT % (util.ROOT, str(tmp_path)))
    rc = ydist.launch_ranks([sys.executable, str(script)], 2, timeout=120)
    assert rc == 0
    assert (tmp_path / 'rank0.txt').read_text() == '2 0 3' and (tmp_path / 'rank1.txt').read_text() == '2 1 3'


def test_launch_ranks_reports_a_failing_rank_and_stops_the_rest(tmp_path):
    import sys
    script = tmp_path / 'fail.py'
    script.write_text('import os, sys, time\nif os.environ["RANK"] == "1":\n    sys.exit(3)\ntime.sleep(60)\n')
    import time
    t0 = time.time()
    assert ydist.launch_ranks([sys.executable, str(script)], 2, timeout=120) == 3
    assert time.time() - t0 < 30


def _forward_nms(model, x):
    with torch.no_grad():
        z, _ = model(x)
    return _pad(yu.non_max_suppression(z, 0.05, 0.6))


def _sharded_model_worker(rank, world, port, q):
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank), MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port))
    torch.set_num_threads(2)
    r, w, _ = ydist.init_from_env(backend='gloo')
    model = util.build_model('yolov5s', seed=0)
    util.calibrate(model, seed=1, size=64, head_gain=2.0)
    model.eval()
    x = torch.rand(6, 3, 64, 64, generator=torch.Generator().manual_seed(3))   # the global batch, the same on every rank
    lo, hi = ydist.shard_bounds(x.shape[0], r, w)
    det_all, cnt_all = ydist.gather_detections(*_forward_nms(model, x[lo:hi]))
    if r == 0:
        det_one, cnt_one = _forward_nms(model, x)
        q.put((det_all.numpy(), cnt_all.numpy(), det_one.numpy(), cnt_one.numpy()))
    dist.barrier()
    dist.destroy_process_group()


def test_global_batch_sharded_over_two_ranks_detects_what_one_process_detects():
    """bench.py --global-batch: every rank runs forward + NMS on its contiguous slice of the batch and the gather puts
    the slices back in image order; the result is the single-process result of the whole batch (same counts, boxes equal
    up to the CPU convolution's batch-size-dependent summation order)."""
    ctx = mp.get_context('spawn')
    q = ctx.SimpleQueue()
    port = _free_port()
    procs = [ctx.Process(target=_sharded_model_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    det_all, cnt_all, det_one, cnt_one = q.get()
    for p in procs:
        p.join(120)
        assert p.exitcode == 0
    assert det_all.shape == det_one.shape == (6, 300, 6)
    assert np.array_equal(cnt_all, cnt_one) and int(cnt_one.min()) > 0
    assert np.allclose(det_all, det_one, rtol=1e-4, atol=1e-3)


def _tune_worker(rank, world, port, q):
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank), MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port))
    from yolo_amd import engine
    ydist.init_from_env(backend='gloo')
    key = (32, 20, 20, 512, 512, 512, 512, 3, 1, False, True, False, 0)
    seen = []

    def warm():  # stands for the forwards whose plans time every variant: rank 0 "tunes", the others must find its choice
        seen.append(dict(engine._TUNED))
        if rank == 0:
            engine._TUNED[key], engine._TUNED_MS[key] = 96, 0.098
    taken = ydist.tune_once(warm)
    q.put((rank, taken, seen[0].get(key), key in engine._REPLAYED))
    dist.barrier()
    dist.destroy_process_group()


def test_tune_once_hands_rank_zeros_choices_to_the_other_ranks():
    ctx = mp.get_context('spawn')
    q = ctx.SimpleQueue()
    port = _free_port()
    procs = [ctx.Process(target=_tune_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    got = dict((r, rest) for r, *rest in (q.get(), q.get()))
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    assert got[0] == [0, None, False]     # rank 0 started from nothing and timed for itself
    assert got[1] == [1, 96, True]        # rank 1 ran warm() with rank 0's choice in place, marked for a trial launch


def test_pin_cpus_gives_ranks_disjoint_shares(monkeypatch):
    have = sorted(os.sched_getaffinity(0))
    if len(have) < 2:
        import pytest
        pytest.skip('one CPU')
    try:
        a = ydist.pin_cpus(0, 2)
        assert a == set(have[:len(have) // 2]) and os.sched_getaffinity(0) == a
        os.sched_setaffinity(0, have)
        b = ydist.pin_cpus(1, 2)
        assert b == set(have[len(have) // 2:2 * (len(have) // 2)]) and not (a & b)
        os.sched_setaffinity(0, have)
        monkeypatch.setenv('Y5_PIN_CPUS', '0')
        assert ydist.pin_cpus(0, 2) is None and sorted(os.sched_getaffinity(0)) == have
        assert ydist.pin_cpus(0, 1) is None
    finally:
        os.sched_setaffinity(0, have)
