"""The autotuner's persisted table (viddet_amd/model.py TuneCache): file round trip, the library-identity check, the
memory-only switch, tolerance of an unreadable file, and - world_size 2 on gloo - every rank adopting rank 0's choice."""
import json
import os
import socket

import torch.distributed as dist
import torch.multiprocessing as mp


def test_table_round_trip_and_identity(tmp_path, monkeypatch):
    from viddet_amd.model import TuneCache
    p = tmp_path / "t" / "tune.json"
    monkeypatch.setenv("VD_TUNE_CACHE", str(p))
    c = TuneCache()
    k1 = ("auto", 64, 26, 26, 256, 26, 26, 1, 9, 512, 1, 3, False, True, True, False, 1, False)
    k2 = ("wgrad", 64, 26, 26, 256, 26, 26, 1, 9, 512, 1, False, 64)
    assert k1 not in c and c.hits == 0
    c[k1] = (64, 5)
    c[k2] = 64
    c.save()
    doc = json.load(open(p))
    assert doc["library"] == TuneCache.library_id() and len(doc["entries"]) == 2
    d = TuneCache()
    assert k1 in d and d[k1] == (64, 5) and d[k2] == 64 and d.tuned == 0 and d.hits == 1
    d.clear()                                            # memory cleared; the file's entries come back on the next lookup
    assert len(d) == 0 and k2 in d
    # a table timed on other kernels is not read
    doc["library"] = "0" * 12
    json.dump(doc, open(p, "w"))
    assert k1 not in TuneCache()
    # an unreadable file is a cold start
    open(p, "w").write("{ not json")
    assert k1 not in TuneCache()
    # memory only
    monkeypatch.setenv("VD_TUNE_CACHE", "off")
    e = TuneCache()
    e[k1] = (0, 2)
    e.save()
    assert e.path() is None and open(p).read() == "{ not json"
    # the default location names the kernel sources the choices were timed on
    monkeypatch.delenv("VD_TUNE_CACHE")
    assert os.path.basename(TuneCache().path()) == "gfx950_%s.json" % TuneCache.library_id()


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, path, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), VD_TUNE_CACHE=path)
    from viddet_amd import dist as vd
    from viddet_amd.model import TuneCache
    vd.init_from_env(backend="gloo")
    try:
        # rank 1 has a table of its own (another node's disk, a stale copy): it must not be read - rank 0's table is the
        # job's table, or rank 0 would enter agree()'s broadcast for ("pre", 1) alone
        if rank == 1:
            os.environ["VD_TUNE_CACHE"] = path + ".rank1"
            json.dump({"library": TuneCache.library_id(), "entries": {"('pre', 1)": [1, 1], "('only1', 1)": 3}}, open(path + ".rank1", "w"))
        else:
            json.dump({"library": TuneCache.library_id(), "entries": {"('pre', 2)": [2, 2]}}, open(path, "w"))
        vd.barrier()
        c = TuneCache()
        assert ("pre", 1) not in c and ("only1", 1) not in c and ("pre", 2) in c and c[("pre", 2)] == (2, 2)
        mine = (64, 5) if rank == 0 else (16, 2)           # the ranks' timings disagree
        got = c.agree(mine)
        c[("k", 1)] = got
        w = c.agree(64 if rank == 0 else 0)
        c[("wgrad", 1)] = w
        c.save()                                           # rank 0 writes
        vd.barrier()
        q.put((rank, got, w, os.path.exists(path)))
    except Exception as e:  # noqa: BLE001
        q.put((rank, "fail: %r" % (e,), None, False))
    finally:
        dist.destroy_process_group()


def test_ranks_adopt_rank0_choice(tmp_path):
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    path = str(tmp_path / "tune.json")
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, path, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=180) for _ in range(world))
    for p in procs:
        p.join(timeout=60)
    assert [r[1] for r in res] == [(64, 5), (64, 5)] and [r[2] for r in res] == [64, 64], res
    assert all(r[3] for r in res)
    doc = json.load(open(path))
    assert doc["entries"] == {"('k', 1)": [64, 5], "('pre', 2)": [2, 2], "('wgrad', 1)": 64}
