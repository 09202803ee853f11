// libnestfit_amd_ring.so: the shared-memory ring alone (client and transport side), no HIP -- the one
// library a sampler process loads (nfa_ring.h; declarations in include/nestfit_amd.h).
#define NFA_RING_STANDALONE
#include <cstddef>
#include <vector>
#include "nfa_ring.h"
