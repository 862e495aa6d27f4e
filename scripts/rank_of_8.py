"""What ONE rank of a row-sharded item-graph build executes, timed on one GPU: record, then replay.

No 8-GPU node has been available to this project, so the 8-rank figure of DESIGN.md section 6 was a model put together from
one-GPU kernel times.  This script measures the compute of a rank instead:

  record   `python -m torch.distributed.run --nproc-per-node W scripts/rank_of_8.py record --rows N [--rank R]`
           W processes share GPU 0 (gloo for the barriers) and run as_build_laplacian_sharded over a transport that moves every
           collective through files in /dev/shm (one collective at a time, so the host never holds more than one of them).  Rank R
           keeps what it RECEIVED in every collective -- the sequence of (operation, offsets, bytes) -- under --dir.
  replay   `python scripts/rank_of_8.py replay --rows N [--rank R]`
           one process is rank R of W: the same call on the same shard, over a transport that answers every collective with the
           recorded bytes, already resident in HBM, by device-to-device copies ON THE STREAM the library hands it (so the gather of
           the f64 rows still runs on its side stream under the join, as RCCL's would).  Nothing is sent: the wall time of the call
           is exactly the work rank R executes, with an exchange that costs a local copy.  Checked: the row block it returns has
           the entry count the recorded run returned, and (option --check) the bytes of the one-GPU build's rows [lo, hi).

Output of replay: one JSON object -- wall ms per call (median of --reps), the stage profile, the bytes received per collective
and in total (what xGMI has to deliver to this rank), the one-GPU build of all N rows beside it.
"""
import argparse
import ctypes as C
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "arrowspace-rs_amd"))

import numpy as np  # noqa: E402
import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402


def parse():
    p = argparse.ArgumentParser()
    p.add_argument("mode", choices=["record", "replay"])
    p.add_argument("--rows", dest="n", type=int, default=4_000_000)
    p.add_argument("--dim", type=int, default=128)
    p.add_argument("--world", type=int, default=8)
    p.add_argument("--rank", type=int, default=3, help="the rank whose receptions are recorded / replayed")
    p.add_argument("--dir", default="/dev/shm/as_rank_of_8")
    p.add_argument("--reps", type=int, default=5)
    p.add_argument("--opt", action="append", default=[], help="replay: context option name=value (repeatable)")
    p.add_argument("--check", action="store_true", help="replay: compare the block with rows [lo, hi) of the one-GPU build")
    p.add_argument("--rank-only", action="store_true", help="replay: no one-GPU build beside the rank's (a kernel trace then ends with the rank's build)")
    p.add_argument("--out", default=None)
    return p.parse_args()


def hip_runtime():
    hip = C.CDLL("libamdhip64.so")
    hip.hipMemcpy.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int]
    hip.hipMemcpyAsync.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int, C.c_void_p]
    hip.hipStreamSynchronize.argtypes = [C.c_void_p]
    return hip


def record_transport(ctx, A, L, rank, world, xdir, keep):
    """Collectives through files in /dev/shm; `keep` (a list, or None): what this rank received, in order."""
    hip = hip_runtime()
    seq = [0]

    def d2h(src, nbytes):
        out = np.empty(int(nbytes), dtype=np.uint8)
        if nbytes:
            assert hip.hipMemcpy(out.ctypes.data, src, int(nbytes), 2) == 0
        return out

    def h2d(dst, arr):
        if arr.size:
            assert hip.hipMemcpy(dst, arr.ctypes.data, arr.size, 1) == 0

    def path(*parts):
        return os.path.join(xdir, "x_" + "_".join(str(p) for p in parts) + ".bin")

    def all_gather_v(user, send, send_bytes, recv, recv_bytes, recv_offsets, stream):
        try:
            hip.hipStreamSynchronize(stream)
            s = seq[0]
            seq[0] += 1
            d2h(send, send_bytes).tofile(path(s, rank))
            dist.barrier()
            rec = {"op": "all_gather_v", "seq": s, "parts": []}
            for r in range(world):
                part = np.fromfile(path(s, r), dtype=np.uint8)
                assert part.size == recv_bytes[r], (part.size, recv_bytes[r])
                h2d((recv or 0) + recv_offsets[r], part)
                if keep is not None:
                    rec["parts"].append((int(recv_offsets[r]), part))
            if keep is not None:
                keep.append(rec)
            dist.barrier()
            os.remove(path(s, rank))
            return 0
        except Exception as e:  # noqa: BLE001 -- a Python exception must not cross the C frame
            sys.stderr.write("record transport: %r\n" % (e,))
            return 9

    def all_to_all_v(user, send, send_bytes, send_offsets, recv, recv_bytes, recv_offsets, stream):
        try:
            hip.hipStreamSynchronize(stream)
            s = seq[0]
            seq[0] += 1
            for p in range(world):
                d2h((send or 0) + send_offsets[p], send_bytes[p]).tofile(path(s, rank, p))
            dist.barrier()
            rec = {"op": "all_to_all_v", "seq": s, "parts": []}
            for p in range(world):
                part = np.fromfile(path(s, p, rank), dtype=np.uint8)
                assert part.size == recv_bytes[p], (part.size, recv_bytes[p])
                h2d((recv or 0) + recv_offsets[p], part)
                if keep is not None:
                    rec["parts"].append((int(recv_offsets[p]), part))
            if keep is not None:
                keep.append(rec)
            dist.barrier()
            for p in range(world):
                os.remove(path(s, rank, p))
            return 0
        except Exception as e:  # noqa: BLE001
            sys.stderr.write("record transport: %r\n" % (e,))
            return 9

    def all_reduce_sum_u64(user, buf, count, stream):
        try:
            hip.hipStreamSynchronize(stream)
            s = seq[0]
            seq[0] += 1
            d2h(buf, 8 * count).tofile(path(s, rank))
            dist.barrier()
            total = np.zeros(int(count), dtype=np.uint64)
            for r in range(world):
                total += np.fromfile(path(s, r), dtype=np.uint64)
            h2d(buf, total.view(np.uint8))
            if keep is not None:
                keep.append({"op": "all_reduce_sum_u64", "seq": s, "parts": [(0, total.view(np.uint8).copy())]})
            dist.barrier()
            os.remove(path(s, rank))
            return 0
        except Exception as e:  # noqa: BLE001
            sys.stderr.write("record transport: %r\n" % (e,))
            return 9

    cbs = (L.AllGatherV(all_gather_v), L.AllToAllV(all_to_all_v), L.AllReduceSumU64(all_reduce_sum_u64))
    ops = L.CommOpsC(None, *cbs)
    h = C.c_void_p()
    ctx.check(ctx.lib.as_comm_create_custom(ctx.h, C.byref(ops), rank, world, C.byref(h)))
    return h, (cbs, ops, hip)


class Replay:
    """Answers every collective with the recorded bytes (device-resident) on the stream it is given."""

    def __init__(self, records, dev):
        self.hip = hip_runtime()
        self.records = []
        self.total_bytes = 0
        for rec in records:
            parts = []
            for off, arr in rec["parts"]:
                t = torch.from_numpy(np.ascontiguousarray(arr)).to(dev)
                parts.append((off, t))
                self.total_bytes += int(arr.size)
            self.records.append({"op": rec["op"], "parts": parts})
        self.at = 0
        self.calls = 0
        self.failed = None

    def rewind(self):
        self.at = 0

    def _answer(self, op, recv, sizes, stream):
        rec = self.records[self.at]
        self.at += 1
        self.calls += 1
        if rec["op"] != op:
            raise RuntimeError("replay: call %d is %s, recorded %s" % (self.at - 1, op, rec["op"]))
        for i, (off, t) in enumerate(rec["parts"]):
            if sizes is not None and int(sizes[i]) != t.numel():
                raise RuntimeError("replay: call %d part %d has %d bytes, recorded %d" % (self.at - 1, i, int(sizes[i]), t.numel()))
            if t.numel():
                rc = self.hip.hipMemcpyAsync((recv or 0) + off, t.data_ptr(), t.numel(), 3, stream)   # device to device, stream-ordered
                if rc != 0:
                    raise RuntimeError("hipMemcpyAsync %d" % rc)

    def make(self, ctx, L, rank, world):
        def all_gather_v(user, send, send_bytes, recv, recv_bytes, recv_offsets, stream):
            try:
                self._answer("all_gather_v", recv, [recv_bytes[r] for r in range(world)], stream)
                return 0
            except Exception as e:  # noqa: BLE001
                self.failed = repr(e)
                return 9

        def all_to_all_v(user, send, send_bytes, send_offsets, recv, recv_bytes, recv_offsets, stream):
            try:
                self._answer("all_to_all_v", recv, [recv_bytes[r] for r in range(world)], stream)
                return 0
            except Exception as e:  # noqa: BLE001
                self.failed = repr(e)
                return 9

        def all_reduce_sum_u64(user, buf, count, stream):
            try:
                self._answer("all_reduce_sum_u64", buf, [8 * count], stream)
                return 0
            except Exception as e:  # noqa: BLE001
                self.failed = repr(e)
                return 9

        cbs = (L.AllGatherV(all_gather_v), L.AllToAllV(all_to_all_v), L.AllReduceSumU64(all_reduce_sum_u64))
        ops = L.CommOpsC(None, *cbs)
        h = C.c_void_p()
        ctx.check(ctx.lib.as_comm_create_custom(ctx.h, C.byref(ops), rank, world, C.byref(h)))
        self._keep = (cbs, ops)
        return h


def main():
    args = parse()
    import arrowspace_amd as A
    from arrowspace_amd import _lib as L
    from arrowspace_amd import distributed as D
    from arrowspace_amd import synth

    n, d, world = args.n, args.dim, args.world
    gp = A.GraphParams(0.5, 16, 16, 2.0, 0.25)
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(dev)
    meta_path = os.path.join(args.dir, "meta.json")

    if args.mode == "record":
        rank = int(os.environ["RANK"])
        assert int(os.environ["WORLD_SIZE"]) == world
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("gloo")
        if rank == 0:
            os.makedirs(args.dir, exist_ok=True)
            for f in os.listdir(args.dir):
                os.remove(os.path.join(args.dir, f))
        dist.barrier()
        ctx = A.Context(0)
        keep = [] if rank == args.rank else None
        h, alive = record_transport(ctx, A, L, rank, world, args.dir, keep)
        comm = D.LibComm(ctx, h, keep=alive)
        lo, hi = comm.shard_range(n)
        X, _ = synth.mixture_device(hi - lo, d, dev, seed=42, row_start=lo)
        t0 = time.perf_counter()
        sc = comm.build_laplacian(X, n, gp)
        ctx.synchronize()
        dt = time.perf_counter() - t0
        if rank == args.rank:
            index = []
            for i, rec in enumerate(keep):
                sizes = []
                for j, (off, arr) in enumerate(rec["parts"]):
                    arr.tofile(os.path.join(args.dir, "rec_%d_%d.bin" % (i, j)))
                    sizes.append([off, int(arr.size)])
                index.append({"op": rec["op"], "parts": sizes})
            with open(meta_path, "w") as f:
                json.dump({"n": n, "dim": d, "world": world, "rank": rank, "lo": lo, "hi": hi, "nnz_total": sc.nnz(),
                           "block_nnz": int(sc.block.nnz()) if sc.block is not None else 0, "records": index,
                           "record_wall_s": dt}, f)
            print("recorded %d collectives for rank %d of %d: %.2f GB received, nnz %d (block %d), build under the file transport %.1f s"
                  % (len(keep), rank, world, sum(p[1] for r in index for p in r["parts"]) / 1e9, sc.nnz(),
                     int(sc.block.nnz()) if sc.block is not None else 0, dt), flush=True)
        dist.barrier()
        dist.destroy_process_group()
        return

    # ---- replay ----------------------------------------------------------------------------------------------------------
    with open(meta_path) as f:
        meta = json.load(f)
    assert (meta["n"], meta["dim"], meta["world"], meta["rank"]) == (n, d, world, args.rank), meta
    rank, lo, hi = args.rank, meta["lo"], meta["hi"]
    records = []
    for i, rec in enumerate(meta["records"]):
        parts = [(off, np.fromfile(os.path.join(args.dir, "rec_%d_%d.bin" % (i, j)), dtype=np.uint8)) for j, (off, size) in enumerate(rec["parts"])]
        records.append({"op": rec["op"], "parts": parts})
    ctx = A.Context(0)
    for kv in args.opt:
        ctx.set_option(kv.split("=")[0], int(kv.split("=")[1]))
    rp = Replay(records, dev)
    del records
    comm = D.LibComm(ctx, rp.make(ctx, L, rank, world))
    assert comm.shard_range(n) == (lo, hi)
    X, _ = synth.mixture_device(hi - lo, d, dev, seed=42, row_start=lo)
    each, prof = [], {}
    for it in range(args.reps + 1):
        rp.rewind()
        if it == args.reps:
            ctx.profile(True)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        sc = comm.build_laplacian(X, n, gp)
        ctx.synchronize()
        dt = time.perf_counter() - t0
        if rp.failed:
            raise SystemExit("replay failed: " + rp.failed)
        assert sc.nnz() == meta["nnz_total"] and int(sc.block.nnz()) == meta["block_nnz"], (sc.nnz(), meta["nnz_total"])
        if it:
            each.append(dt * 1e3)
        if it == args.reps:
            prof = {k: round(v[0], 4) for k, v in ctx.profile_all().items() if v[1]}
            ctx.profile(False)
        block = sc.block
    stats = {k: ctx.stat(k) for k in ("sharded_by_positions", "sharded_prepare_by_rows", "sharded_rows_late", "sharded_foreign_slots",
                                      "sharded_listed_slots", "knn_join_candidates", "knn_uncertified_rows", "knn_mfma_join_f16_blocks",
                                      "knn_mfma_join_f16_blocks_listed", "knn_mfma_join_f16_blocks_pretest", "knn_join_prefilter_blocks",
                                      "knn_join_prefilter_blocks_flagged", "knn_join_cone_tiles_kept_e6")}
    by_op = {}
    for rec in meta["records"]:
        e = by_op.setdefault(rec["op"], {"calls": 0, "bytes": 0})
        e["calls"] += 1
        e["bytes"] += sum(p[1] for p in rec["parts"])
    own = sum(rec["parts"][rank][1] for rec in meta["records"] if rec["op"] != "all_reduce_sum_u64" and len(rec["parts"]) == world)
    largest = sorted(((sum(p[1] for p in rec["parts"]), i, rec["op"]) for i, rec in enumerate(meta["records"])), reverse=True)[:6]
    out = {"what": "rank %d of %d of as_build_laplacian_sharded at %d x %d, replayed on one GPU: every collective answered with the recorded "
                   "bytes by a device-to-device copy on the library's stream (nothing is sent; the exchange costs a local copy)" % (rank, world, n, d),
           "n": n, "dim": d, "world": world, "rank": rank, "rows_of_the_rank": hi - lo,
           "rank_ms": sorted(each)[len(each) // 2], "rank_ms_each": [round(e, 3) for e in each],
           "collectives": len(meta["records"]), "received_bytes": rp.total_bytes, "received_bytes_from_peers": rp.total_bytes - own,
           "received_by_operation": by_op,
           "largest_collectives": [{"call": i, "op": op, "bytes": b} for b, i, op in largest],
           "xgmi_floor_ms_at_7x153GBs": (rp.total_bytes - own) / (7 * 153e9) * 1e3,
           "block_nnz": meta["block_nnz"], "nnz_total": meta["nnz_total"], "stats": stats, "stage_ms": prof}
    if not args.rank_only:
        # the one-GPU build of all N rows beside it (time, and with --check the block's bytes)
        Xall, _ = synth.mixture_device(n, d, dev, seed=42)
        A.build_laplacian_matrix(Xall, gp, ctx=ctx)
        ctx.synchronize()
        one = []
        for _ in range(3):
            t0 = time.perf_counter()
            gl = A.build_laplacian_matrix(Xall, gp, ctx=ctx)
            ctx.synchronize()
            one.append((time.perf_counter() - t0) * 1e3)
        out["one_gpu_ms"] = sorted(one)[1]
        out["one_gpu_ms_each"] = [round(e, 3) for e in one]
        out["one_gpu_over_rank"] = out["one_gpu_ms"] / out["rank_ms"]
        assert gl.matrix.nnz() == meta["nnz_total"], (gl.matrix.nnz(), meta["nnz_total"])
        if args.check:
            ip, ix, dv = gl.matrix.to_host()
            bp, bx, bv = block.to_host()
            a, b = int(ip[lo]), int(ip[hi])
            assert np.array_equal(bp.astype(np.int64) - int(bp[0]), ip[lo:hi + 1].astype(np.int64) - a)
            assert np.array_equal(bx, ix[a:b]) and np.array_equal(bv, dv[a:b])
            out["block_equals_rows_of_the_one_gpu_build"] = True
    text = json.dumps(out, indent=1)
    print(text)
    if args.out:
        with open(args.out, "w") as f:
            f.write(text + "\n")


if __name__ == "__main__":
    main()
