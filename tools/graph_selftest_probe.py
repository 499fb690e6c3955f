import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
if os.environ.get("LATE"): torch.zeros(1, device="cuda")          # runtime initialised with the variable as the shell set it
import pnode_amd
from pnode_amd import _graphcheck
torch.zeros(1, device="cuda"); torch.cuda.synchronize()
t0 = time.time(); ok = _graphcheck.replay_is_sound(torch.device("cuda:0")); dt = time.time() - t0
print("env", os.environ.get("DEBUG_CLR_GRAPH_PACKET_CAPTURE"), "SAFE flag", pnode_amd.GRAPH_REPLAY_SAFE, "self-test sound:", ok, "%.2f s" % dt, "mem after %.1f MB" % (torch.cuda.memory_allocated() / 1e6))
