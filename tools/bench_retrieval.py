"""Retrieval-only benchmark (10k x 200k x 512 top-10 + protocol metrics), for rocprofv3 runs."""
import os, sys, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
print(json.dumps(bench.retrieval_bench(torch.device('cuda:0'))))
