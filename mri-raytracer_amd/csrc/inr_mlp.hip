// INR forward — placeholder translation unit (replaced by the MFMA kernel); entry points report
// MRIRT_ERR_ARG so a caller fails loudly instead of reading garbage.
#include "mrirt_host.h"
extern "C" int64_t mrirt_inr_pack_bytes(const MrirtInrDesc*) { return 0; }
extern "C" int mrirt_inr_pack_weights(const MrirtInrDesc*, const float*, void*, void*) { return MRIRT_ERR_ARG; }
extern "C" int mrirt_inr_forward(const MrirtInrDesc*, const float*, const float*, int64_t, float*, int16_t*, void*) { return MRIRT_ERR_ARG; }
extern "C" int mrirt_inr_predict_volume(const MrirtInrDesc*, const float*, const uint32_t*, int16_t*, void*) { return MRIRT_ERR_ARG; }
