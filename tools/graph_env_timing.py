import os, sys, subprocess
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
when = sys.argv[1]
if when == "before_import":
    os.environ["DEBUG_CLR_GRAPH_PACKET_CAPTURE"] = "0"
import torch
if when == "after_import":
    os.environ["DEBUG_CLR_GRAPH_PACKET_CAPTURE"] = "0"
if when == "after_devcount":
    torch.cuda.device_count()
    os.environ["DEBUG_CLR_GRAPH_PACKET_CAPTURE"] = "0"
if when == "after_init":
    torch.zeros(1, device="cuda")
    os.environ["DEBUG_CLR_GRAPH_PACKET_CAPTURE"] = "0"
os.environ["NV"] = "2"
sys.argv = ["x"]
exec(open(os.path.join(ROOT, "tools", "graph_sum_repro2.py")).read())
