"""Where does a single-peer RCCL all-to-all message start to arrive wrong?  One-rank nccl group on one GPU, all_to_all_single of
a float64 tensor to itself, sizes around 1, 2 and 4 GiB; prints the first wrong byte offset.  (shenqi_amd/dist.py keeps every
peer message under 1 GiB; this is the evidence for that limit.)"""
import os
import torch
import torch.distributed as dist

os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
os.environ.setdefault("MASTER_PORT", "29544")
torch.cuda.set_device(0)
dev = torch.device("cuda", 0)
dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
import sys
DT = {"f64": torch.float64, "u8": torch.uint8, "f32": torch.float32}
for spec in (sys.argv[1:] or ["f64:0.9", "f64:0.999", "f64:1.001", "f64:1.5", "u8:0.999", "u8:1.001", "f32:1.001", "f64:3.6"]):
    dtn, gib = spec.split(":"); gib = float(gib); dt = DT[dtn]
    esz = torch.empty(0, dtype=dt).element_size()
    n = int(gib * (1 << 30) / esz)
    s = (torch.arange(n, dtype=torch.int64, device=dev) % 251 + 1).to(dt)
    r = torch.zeros_like(s)
    dist.all_to_all_single(r, s)
    torch.cuda.synchronize()
    bad = (r != s)
    nbad = int(bad.sum())
    first = int(torch.nonzero(bad)[0]) * esz if nbad else -1
    unwritten = int((r[bad] == 0).sum()) if nbad else 0
    print("%s %.3f GiB (%d bytes): %d wrong elements%s" % (dtn, gib, n * esz, nbad, "" if nbad == 0 else
          ", first at byte offset %d (%.4f of the message), %d of them never written" % (first, first / (n * esz), unwritten)), flush=True)
    del s, r, bad
dist.destroy_process_group()
