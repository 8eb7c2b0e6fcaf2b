// Attention dispatcher (the MFMA flash kernel is added below as it lands).
#include "ops.h"

int attention_scalar(const AttentionArgs& a, hipStream_t stream);

int attention(const AttentionArgs& a, hipStream_t stream) {
  return attention_scalar(a, stream);
}
