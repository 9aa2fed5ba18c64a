"""Developer: N encodes of ONE question with the default provider (for rocprofv3 --kernel-trace --stats)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from rag_dpo_amd import synth
from rag_dpo_amd.embedding_provider import EmbeddingProvider
p = EmbeddingProvider(model_name="random-init:xlm-roberta-large", device="cuda:0", dtype=torch.float16, batch_size=64).load()
if len(sys.argv) > 2 and sys.argv[2] == "eager":
    p._packed.graphs = False
q = synth.query_texts(4)
for _ in range(int(sys.argv[1]) if len(sys.argv) > 1 else 50):
    p.embed_device(q[:1]); torch.cuda.synchronize()
