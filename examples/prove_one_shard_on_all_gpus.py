"""One shard proved by all GPUs of a node together: every rank ends with the words one GPU would have produced.

    torchrun --nnodes=1 --nproc-per-node 8 --master-addr 127.0.0.1 examples/prove_one_shard_on_all_gpus.py [n]
    torchrun --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 examples/prove_one_shard_on_all_gpus.py --share-one-gpu [n]

The reference proves an execution below 2^22 rows as ONE shard inside one process (/root/reference/src/lair/execute.rs:231-241,
/root/reference/benches/fib.rs:124): `machine.prove` there is `SplitProver.prove` here, with a communicator.  With `--share-one-gpu` the
ranks share device 0 and the collectives go through host memory over gloo (RCCL refuses two ranks on one device): a rehearsal, not
a speed-up."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

import torch
import torch.distributed as dist

import lurk_amd
from lurk_amd import lair, prover, split
from prove_and_verify import SOURCE  # the fib program of the one-GPU example


def main(n: int, share: bool):
    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    device = 0 if share else int(os.environ.get("LOCAL_RANK", rank))
    torch.cuda.set_device(device)
    dist.init_process_group("gloo" if share else "nccl")
    # every rank executes (or receives: lurk_amd.shards.scatter_prepared) the shard's kernel inputs
    top = lair.Toplevel(SOURCE)
    record = lair.QueryRecord(top)
    out = top.execute_by_name("fib", [n], record)
    public_values = record.expect_public_values()
    with lurk_amd.Context(device) as ctx:
        machine = prover.Machine(ctx, top, "fib", len(public_values))
        machine.setup()
        prepared = machine.prepare_shard(lair.Shard.new(record))
        comm_c = None
        if share:
            comm = split.TorchSplitComm(ctx)
        else:
            from lurk_amd.comm import Comm

            comm_c = Comm.from_process_group(ctx)          # this library's RCCL communicator (csrc/comm.cpp)
            comm = split.RcclSplitComm(ctx, comm_c)        # grouped ncclSend / ncclRecv pairs: device buffers stay on the devices
            bad = comm.selftest()                          # the four collectives once, with known words
            if bad is not None:
                raise SystemExit(f"rank {rank}: the RCCL carrier's self-test failed: {bad}")
        sp = split.SplitProver(machine, comm, min_log_n=max(8, world.bit_length() - 1))  # chips of at least 2^8 rows are cut across the ranks
        sp.setup()
        blocks = sp.run_prepared_blocks(prepared)          # a cut chip's trace: this rank's block of rows only
        ctx.sync()
        dist.barrier()
        t0 = time.perf_counter()
        words, root = sp.prove(blocks, public_values, 100, 16, row_blocks=True)
        ctx.sync()
        t_prove = time.perf_counter() - t0
        # every rank holds the whole proof; any of them (or nobody with a GPU) can verify it
        ok = bool(machine.verify([words])) if rank == 0 else True
        crc = torch.tensor([int(words.astype("uint32").sum() % (1 << 31))], dtype=torch.int64)
        crcs = [torch.zeros_like(crc) for _ in range(world)]
        dist.all_gather(crcs, crc.cuda() if not share else crc)
        same = len({int(c.item()) for c in crcs}) == 1
        if rank == 0:
            print(f"fib({n}) mod p = {out[0]}: one shard of {record.num_func_queries(0)} rows proved by {world} ranks in {t_prove * 1e3:.1f} ms, "
                  f"{len(words)} proof words, the same on every rank: {same}, verified: {ok}")
        dist.barrier()
        sp.close()
        if comm_c is not None:
            comm_c.close()
        machine.close()
    dist.destroy_process_group()
    if not (ok and same):
        raise SystemExit(1)


if __name__ == "__main__":
    args = [a for a in sys.argv[1:] if a != "--share-one-gpu"]
    main(int(args[0]) if args else 3000, "--share-one-gpu" in sys.argv)
