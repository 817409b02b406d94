"""world_size-2 test of the long-file sharding on CPU (gloo).  The sharding / halo / gather logic is
host code shared with the GPU path; here the per-rank separation is played by the CPU oracle (test
infrastructure), on the GPU box it is Separator.separate."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from deepconvsep_amd.dist import plan_long_file, shard_ranges

N, HOP, TC, OV = 256, 64, 30, 25


def test_shard_ranges_cover_everything_once():
    for n in (0, 1, 7, 32, 255, 4096):
        for w in (1, 2, 3, 8):
            r = shard_ranges(n, w)
            assert r[0][0] == 0 and r[-1][1] == n and all(a[1] == b[0] for a, b in zip(r, r[1:]))
            assert max(h - l for l, h in r) - min(h - l for l, h in r) <= 1


def test_plan_is_a_partition_on_the_tile_grid():
    L = 48000
    for w in (1, 2, 4, 8):
        plan = plan_long_file(L, w, N, HOP, TC, OV)
        assert plan[0]['s0'] == 0 and plan[-1]['s1'] == L
        for a, b in zip(plan, plan[1:]):
            assert a['s1'] == b['s0']
        for p in plan:
            assert p['a0'] <= p['s0'] <= p['s1'] <= p['a1']
            assert p['a0'] % ((TC - OV) * HOP) == 0          # segment tiles coincide with the global tiles


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def _worker(rank, world, port, audio, params, ret):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.set_num_threads(2)
    from deepconvsep_amd.dist import gather_batches, separate_long_file
    from oracle import pipeline

    def separate_fn(seg):
        return pipeline.separate("dsd", params, seg, 0.3, TC, OV, 32, N, HOP, np.hanning)

    out = separate_long_file(separate_fn, audio, N, HOP, TC, OV)
    g = gather_batches(torch.full((2, 5), float(rank)))
    if rank == 0:
        ret["pcm"] = out.numpy()
        ret["gathered"] = g.numpy()
    dist.destroy_process_group()


def test_two_ranks_reproduce_the_single_process_result():
    from deepconvsep_amd.synth import synth_audio, synth_params
    from oracle import pipeline
    params = synth_params("dsd", TC, N // 2 + 1, seed=2)
    audio = synth_audio(40000, seed=3)
    want = pipeline.separate("dsd", params, audio, 0.3, TC, OV, 32, N, HOP, np.hanning)
    mgr = mp.Manager()
    ret = mgr.dict()
    mp.spawn(_worker, args=(2, _free_port(), audio, params, ret), nprocs=2, join=True)
    got = ret["pcm"]
    assert got.shape == want.shape
    assert np.max(np.abs(got - want)) < 1e-9
    assert np.array_equal(ret["gathered"], np.repeat(np.array([0.0, 0.0, 1.0, 1.0])[:, None], 5, axis=1))
