"""Developer: N encodes of BASELINE config 5's 1024 questions with the default provider (for rocprofv3 --kernel-trace --stats)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from rag_dpo_amd import synth
from rag_dpo_amd.embedding_provider import EmbeddingProvider
p = EmbeddingProvider(model_name="random-init:xlm-roberta-large", device="cuda:0", dtype=torch.float16, batch_size=1024).load()
texts = synth.query_texts(1024)
n = int(sys.argv[1]) if len(sys.argv) > 1 else 20
for _ in range(n):
    p.embed_device(texts)
torch.cuda.synchronize()
print("encodes", n, p.last_encode_stats)
