from .activation import MojoSwiGLU
from .attention import MojoPagedDecodeGQA, MojoPagedPrefillGQA
from .compute_with_comm import MojoAllGatherGemm, MojoGemmAll2All, MojoGemmAllReduce, MojoGemmReduceScatter
from .gemm import MojoGemm, MojoGroupGemm, MojoQuantGemm
from .kv_cache import MojoStorePagedKVCache, MojoStorePagedMLAKVCache, build_paged_kv_chunk_metadata
from .mla import MojoPagedDecodeMLA, MojoPagedPrefillMLA
from .mlp import MojoSwiGLUMLP
from .moe import MojoExperts, MojoMoE, MojoMoECombine, MojoMoEDispatch, MojoMoEGating
from .normalization import MojoResidualAddRMSNorm, MojoRMSNorm, MojoRMSNormInplace
from .position_embedding import MojoApplyRoPE, MojoRotaryEmbedding
from .quantize import MojoDynamicQuant, MojoResidualAddRMSNormQuant

__all__ = [
    "MojoSwiGLU", "MojoPagedDecodeGQA", "MojoPagedPrefillGQA", "MojoAllGatherGemm", "MojoGemmAll2All",
    "MojoGemmAllReduce", "MojoGemmReduceScatter", "MojoGroupGemm", "MojoQuantGemm", "MojoStorePagedKVCache",
    "build_paged_kv_chunk_metadata", "MojoPagedDecodeMLA", "MojoPagedPrefillMLA", "MojoResidualAddRMSNorm",
    "MojoRMSNorm", "MojoRMSNormInplace", "MojoApplyRoPE", "MojoRotaryEmbedding", "MojoMoEGating", "MojoMoEDispatch", "MojoExperts",
    "MojoMoECombine", "MojoMoE", "MojoDynamicQuant", "MojoResidualAddRMSNormQuant", "MojoStorePagedMLAKVCache",
    "MojoGemm", "MojoSwiGLUMLP",
]
