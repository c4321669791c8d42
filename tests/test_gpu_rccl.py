"""The `nccl` (= RCCL) branch of the multi-GPU helpers, executed on hardware.  A one-GPU box cannot hold two RCCL ranks (one rank
per device), so this is a WORLD = 1 group: communicator set-up in this environment (HSA_ENABLE_IPC_MODE_LEGACY=0), the byte-view
`all_gather_into_tensor` the gallery build and the sharded ranking issue (bf16 / int32 / fp32 device blocks seen by RCCL as uint8),
the object collectives of the tuner-plan broadcast and the ragged gathers, and bench.py's pre-flight on its RCCL path (device-side
all-reduce of the timing, payload check, rate arithmetic).  What it cannot show is bytes crossing xGMI: the 1 / 2 / 4 / 8 curve is the
driver's to measure."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

SCRIPT = r"""
import json, os, sys
sys.path.insert(0, os.environ["FERN_ROOT"])
import torch
import torch.distributed as dist
os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=os.environ["FERN_PORT"], RANK="0", WORLD_SIZE="1", LOCAL_RANK="0")
torch.cuda.set_device(0)
dev = torch.device("cuda", 0)
dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
from fashionern_aaai2024_amd import distributed as fd
import bench
assert dist.get_backend() == "nccl" and fd.world_info() == (0, 1)
out = {}
for name, x in (("bf16", torch.randn(1000, 64, device=dev).bfloat16()), ("int32", torch.arange(50 * 9, dtype=torch.int32, device=dev).view(50, 9)),
                ("f32", torch.randn(7, 513, device=dev))):
    y = torch.empty_like(x)
    fd._all_gather_into(y, x)                      # RCCL all_gather_into_tensor on the uint8 view
    torch.cuda.synchronize()
    out[name] = bool(torch.equal(x, y))
box = ["f32 12608 2304 768 0 0 20 10880 12608\n"]
dist.broadcast_object_list(box, src=0)
meta = [None]
dist.all_gather_object(meta, (3, ["a", "b", "c"]))
out["objects"] = box[0].startswith("f32 12608") and meta[0] == (3, ["a", "b", "c"])
info = bench.multi_gpu_preflight(torch, dist, fd, 0, 1, 0, dev, "nccl")
out["preflight"] = {"rccl_world": info["rccl_world"], "backend": info["backend"], "payload_ok": info["all_gather_64MiB_per_rank"]["payload_ok"],
                    "ms": info["all_gather_64MiB_per_rank"]["ms"], "device": info["ranks"][0]["name"]}
t = torch.tensor([1.5], dtype=torch.float64, device=dev)
dist.all_reduce(t, op=dist.ReduceOp.MAX)
out["all_reduce"] = t.item()
# round 6: the prepared gallery gathered shard-wise (distributed.all_gather_prepared) on the real engine: rows, bf16 copy and the four
# norms equal those of preparing the whole gallery, also into a pre-allocated store; its MAX-reduce of the norms on the device
from fashionern_aaai2024_amd.engine import FernEngine, PreparedGallery
eng = FernEngine(dev)
g = torch.nn.functional.normalize(torch.randn(1000, 512, device=dev), dim=-1)
whole = eng.prepare_gallery(g)
pg = fd.all_gather_prepared(eng, g, 1000)
store = PreparedGallery(torch.empty_like(g), torch.empty(g.shape, dtype=torch.bfloat16, device=dev), torch.zeros(4, device=dev))
pg2 = fd.all_gather_prepared(eng, g, 1000, out=store)
m = fd._all_reduce_max(whole.meta.clone())
q = torch.nn.functional.normalize(torch.randn(8, 512, device=dev), dim=-1)
s0, i0 = eng.sim_topk(q, whole, 50)
s1, i1 = eng.sim_topk(q, pg2, 50)
torch.cuda.synchronize()
out["prepared"] = bool(torch.equal(pg.f32, whole.f32) and torch.equal(pg.bf16.view(torch.int16), whole.bf16.view(torch.int16)) and torch.equal(pg.meta, whole.meta)
                       and torch.equal(pg2.bf16.view(torch.int16), whole.bf16.view(torch.int16)) and torch.equal(pg2.meta, whole.meta) and torch.equal(m, whole.meta)
                       and pg2.f32.data_ptr() == store.f32.data_ptr() and torch.equal(i0, i1) and torch.equal(s0, s1))
eng.close()
dist.barrier()
dist.destroy_process_group()
print("RESULT " + json.dumps(out))
"""


@pytest.mark.gpu
def test_rccl_world1_group_runs_the_byte_view_all_gather_the_object_collectives_and_the_preflight(tmp_path):
    import socket
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    env = dict(os.environ, FERN_ROOT=ROOT, FERN_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
    r = subprocess.run([sys.executable, "-c", SCRIPT], cwd=ROOT, env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, (r.stdout[-2000:], r.stderr[-3000:])
    line = [l for l in r.stdout.splitlines() if l.startswith("RESULT ")][-1]
    out = json.loads(line[len("RESULT "):])
    assert out["bf16"] and out["int32"] and out["f32"] and out["objects"]
    assert out["preflight"]["rccl_world"] == 1 and out["preflight"]["backend"] == "nccl" and out["preflight"]["payload_ok"]
    assert out["preflight"]["ms"] > 0 and out["all_reduce"] == 1.5
    assert out["prepared"], "all_gather_prepared differs from preparing the whole gallery"
