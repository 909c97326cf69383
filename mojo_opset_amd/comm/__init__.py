from .pipelines import (GemmEngine, all_gather_gemm, gemm_all2all, gemm_all_reduce, gemm_reduce_scatter,
                        plan_row_chunks)

__all__ = ["GemmEngine", "all_gather_gemm", "gemm_all2all", "gemm_all_reduce", "gemm_reduce_scatter",
           "plan_row_chunks"]
