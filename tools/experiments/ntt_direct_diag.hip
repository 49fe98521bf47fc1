// Diagnostic builds of the direct NTT passes (csrc/ntt_direct.hip): this translation unit sets the hook points of that file from
// -DDIRECT_DIAG_* flags and includes it. NEVER the product — the results of most variants are wrong by design; they answer "what
// does this part of the pass cost" (tools/gpu_runs/ntt_direct_variants.sh, profiles/r03_ntt_direct_diagnostic_variants.jsonl):
//   DIRECT_DIAG_SAME_LOADS   every tile loads tile 0's addresses (served by L2)     -> what the load latency costs
//   DIRECT_DIAG_SAME_STORES  every tile stores to tile 0's addresses                 -> what the store traffic costs
//   DIRECT_DIAG_NO_BARRIER   the two workgroup barriers of a tile are dropped        -> what waiting for the slowest wave costs
//   DIRECT_DIAG_TAIL_FRONT   the sixteen tail steps run before the first rounds      -> what spreading the stores buys
//   DIRECT_DIAG_NT_{LOAD,STORE}_{COL,ROW}   nontemporal loads / stores in either pass
#define DIRECT_DIAG_HOOKS
#ifdef DIRECT_DIAG_NO_BARRIER
#define DIRECT_TILE_BARRIER() tile_sync<64>()
#else
#define DIRECT_TILE_BARRIER() lds_barrier()
#endif
#ifdef DIRECT_DIAG_SAME_LOADS
#define DIRECT_LOAD_TILE(t) 0u
#else
#define DIRECT_LOAD_TILE(t) (t)
#endif
#ifdef DIRECT_DIAG_SAME_STORES
#define DIRECT_STORE_TILE(t) 0u
#define DIRECT_DIAG_SAME_STORES_ON 1
#else
#define DIRECT_STORE_TILE(t) (t)
#define DIRECT_DIAG_SAME_STORES_ON 0
#endif
#ifdef DIRECT_DIAG_TAIL_FRONT
#define DIRECT_DIAG_TAIL_FRONT_ON 1
#else
#define DIRECT_DIAG_TAIL_FRONT_ON 0
#endif
#ifdef DIRECT_DIAG_NT_LOAD_COL
#define DIRECT_NT_LOAD_COL true
#else
#define DIRECT_NT_LOAD_COL false
#endif
#ifdef DIRECT_DIAG_NT_STORE_COL
#define DIRECT_NT_STORE_COL true
#else
#define DIRECT_NT_STORE_COL false
#endif
#ifdef DIRECT_DIAG_NT_LOAD_ROW
#define DIRECT_NT_LOAD_ROW true
#else
#define DIRECT_NT_LOAD_ROW false
#endif
#ifdef DIRECT_DIAG_NT_STORE_ROW
#define DIRECT_NT_STORE_ROW true
#else
#define DIRECT_NT_STORE_ROW false
#endif
#include "ntt_direct.hip"
