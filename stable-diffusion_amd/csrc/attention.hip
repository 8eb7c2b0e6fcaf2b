// Attention dispatcher (the MFMA flash kernel is added below as it lands).
#include "ops.h"

int attention_scalar(const AttentionArgs& a, hipStream_t stream);

int attention(const AttentionArgs& a, hipStream_t stream) {
  ProfScope prof(PROF_ATTENTION, stream, 4.0 * a.B * a.heads * (double)a.Nq * a.Nk * a.d, 0.0);
  return attention_scalar(a, stream);
}
