#!/usr/bin/env python3
"""Runs tsvgp_potrf_f64 at M = 1024 a few times per variant (for rocprofv3 --kernel-trace --stats).  usage: dev_potrf_trace.py [flags]"""
import importlib, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
B = importlib.import_module("t-svgp_amd._backend")
estep = importlib.import_module("t-svgp_amd.estep")
eng = estep.EStepEngine(torch.float64, "cuda:0")
eng.potrf_flags = int(sys.argv[1]) if len(sys.argv) > 1 else 0
M = 1024
A = torch.randn(1, M, M, dtype=torch.float64, device="cuda:0")
A = A @ A.transpose(-1, -2) / M + torch.eye(M, dtype=torch.float64, device="cuda:0")
for _ in range(10):
    eng.cholesky(A)
torch.cuda.synchronize()
