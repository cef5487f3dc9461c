"""Rank programs for tests/test_bench_launcher.py: started by bench.launch_ranks the way bench.py's own ranks are (torch.distributed.run,
WORLD_SIZE / RANK in the environment), they go through bench.py's own process-group set-up (bench.init_group: the bounded timeout)
and watchdog (bench.Deadline) over gloo on the CPU, and then misbehave as told:

  die    rank 1 exits with status 17 after the first collective, while rank 0 waits in the next one
  hang   rank 1 never joins the second collective (it sleeps); rank 0's watchdog has a 3 s limit for that stage
  sleep  nobody does anything for ever, and nobody has a watchdog: only the launcher's own deadline ends this
  ok     both ranks finish; rank 0 prints one JSON line
"""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch                      # noqa: E402
import torch.distributed as dist  # noqa: E402

import bench                      # noqa: E402

mode = sys.argv[sys.argv.index("--mode") + 1]
rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
if mode == "sleep":
    time.sleep(3600)
record_fd = os.dup(1)
watchdog = bench.Deadline(rank, record_fd)
watchdog.stage("process group")
bench.init_group("gloo", rank, world)
t = torch.ones(1)
watchdog.stage("set-up")
dist.all_reduce(t)
assert t.item() == world
if mode == "die" and rank == 1:
    os._exit(17)
if mode == "hang" and rank == 1:
    time.sleep(3600)
watchdog.stage("timed loop", 3.0 if mode == "hang" else None)
dist.all_reduce(t)
watchdog.done()
if rank == 0:
    print(json.dumps({"value": t.item()}), flush=True)
dist.destroy_process_group()
