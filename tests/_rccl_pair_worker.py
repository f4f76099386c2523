"""One rank of the two-GPU exchange test (tests/test_gpu_multi.py::test_rccl_allgather_two_ranks): started as a fresh
process per rank, before anything in it has touched a GPU.

    python _rccl_pair_worker.py <id file> <result file>      (RANK, WORLD_SIZE, MASTER_ADDR, MASTER_PORT in the environment)

Rank 0 makes the library's communicator id and hands it over through <id file>; every rank then creates its
dspx communicator on GPU `rank`, gathers a shard through dspx_allgather_f64_dev and -- for comparison -- the same shard
through torch.distributed (RCCL as well), and writes what it found as JSON."""
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main() -> int:
    id_file, result_file = sys.argv[1], sys.argv[2]
    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    import torch
    import torch.distributed as dist

    import dsp_amd

    torch.cuda.set_device(rank)
    dev = torch.device("cuda", rank)
    if rank == 0:
        ident = dsp_amd.Comm.unique_id()
        with open(id_file + ".tmp", "wb") as f:
            f.write(ident)
        os.replace(id_file + ".tmp", id_file)
    else:
        limit = time.monotonic() + 120
        while not os.path.exists(id_file):
            if time.monotonic() > limit:
                raise SystemExit("no communicator id from rank 0")
            time.sleep(0.05)
        with open(id_file, "rb") as f:
            ident = f.read()
    comm = dsp_amd.Comm(ident, world, rank)
    rows, n = 5, 100_003
    mine = torch.empty((rows, n), dtype=torch.float64, device=dev)
    dsp_amd.fill_splitmix(mine, dsp_amd.SEED, rank * rows * n)
    full = comm.allgather(mine)
    torch.cuda.synchronize()
    expect = torch.empty((world * rows, n), dtype=torch.float64, device=dev)
    dsp_amd.fill_splitmix(expect, dsp_amd.SEED, 0)          # rank r's shard is rows [r*rows, (r+1)*rows) of this
    ok_own = bool(torch.equal(full, expect))
    # a call with another device current must be refused, not handed to RCCL
    other = (rank + 1) % world
    torch.cuda.set_device(other)
    refused = False
    try:
        comm.allgather(mine)
    except dsp_amd.DspxError:
        refused = True
    torch.cuda.set_device(rank)
    comm.close()
    dist.init_process_group(backend="nccl", device_id=dev)
    via_torch = dsp_amd.allgather_channels(mine, world * rows)
    torch.cuda.synchronize()
    ok_torch = bool(torch.equal(via_torch, full))
    dist.barrier()
    dist.destroy_process_group()
    with open(result_file, "w") as f:
        json.dump({"rank": rank, "ranks": world, "gathered_equals_expected": ok_own, "equals_torch_gather": ok_torch,
                   "wrong_device_refused": refused}, f)
    return 0 if ok_own and ok_torch and refused else 1


if __name__ == "__main__":
    raise SystemExit(main())
