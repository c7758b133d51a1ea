// behz.hip -- ct x ct (BEHZ) and relinearisation entry points.  (placeholder: implemented next)
#include "../../include/fhe_hip.h"
extern "C" size_t fhe_multiply_scratch_bytes(const fhe_ctx *, uint32_t, uint32_t, uint64_t) { return 0; }
extern "C" int fhe_multiply(const fhe_ctx *, const uint64_t *, uint32_t, const uint64_t *, uint32_t, uint64_t *, uint64_t, void *, size_t, fhe_stream) { return FHE_ERR_PARAM; }
extern "C" int fhe_square(const fhe_ctx *, const uint64_t *, uint32_t, uint64_t *, uint64_t, void *, size_t, fhe_stream) { return FHE_ERR_PARAM; }
extern "C" uint32_t fhe_evk_digits(const fhe_ctx *, uint32_t) { return 0; }
extern "C" int fhe_relinearize(const fhe_ctx *, uint64_t *, uint64_t, uint64_t, const uint64_t *, uint32_t, void *, size_t, fhe_stream) { return FHE_ERR_PARAM; }
extern "C" size_t fhe_relinearize_scratch_bytes(const fhe_ctx *, uint32_t, uint64_t) { return 0; }
