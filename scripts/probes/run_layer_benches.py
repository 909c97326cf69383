import json, sys, torch
import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from benchmarks import extras
dev = torch.device("cuda", 0)
print(json.dumps(extras.bench_decode_layer(dev)))
print(json.dumps(extras.bench_dense_mid_m(dev)))
