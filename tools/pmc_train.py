"""Short training workload for rocprofv3 --pmc passes: two steps of the full 20+4-layer d=1024 model, batch 4 x 499 tokens
(MFMA attention forward + backward, csrc/sgemm.hip in its three operand forms)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from bench import synth_tokens
from shapeformer_amd.gpt import CondTupleGPT
from shapeformer_amd.train import GPTTrainer
dev = torch.device("cuda:0")
g = CondTupleGPT(device=dev)
tr = GPTTrainer(g, lr=1e-5)
c, z = synth_tokens(1000, 4, 200, 300)
for _ in range(2):
    loss = tr.training_step(c, z)
torch.cuda.synchronize()
print("train done", float(loss))
