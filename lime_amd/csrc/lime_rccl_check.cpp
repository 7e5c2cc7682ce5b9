// lime_rccl_check.cpp -- lime_comm.cpp declares the few RCCL types and enum values it uses by hand (so that building the
// library does not drag in rccl.h); this translation unit includes the installed header and pins them.
#if __has_include(<rccl/rccl.h>)
#include <rccl/rccl.h>
static_assert(ncclSum == 0 && ncclMax == 2, "ncclRedOp_t values used by lime_comm.cpp");
static_assert(ncclUint8 == 1 && ncclUint32 == 3 && ncclUint64 == 5, "ncclDataType_t values used by lime_comm.cpp");
static_assert(sizeof(ncclUniqueId) == 128, "ncclUniqueId is LIME_COMM_ID_BYTES long");
#endif
