// beam_state.hip — the generator's per-step bookkeeping on the device.
//
// TextRecognitionGenerator._generate (marie/models/unilm/trocr/generator.py:127-362) interleaves the decoder forward with
// small, branchy list work per sentence: finalize_hypos over the eos candidates among the first `beam` (:225-248), the `beam`
// best live candidates become the next step's rows (:297-343), the finished hypotheses are sorted by score at the end (:362-372).
// Done on the host that work costs a stream drain per step (candidates down, tokens / parents up).  Here one thread per crop
// does it where the candidates already are; the host only watches a "crops remaining" counter, one step late.
// Finished crops keep their rows (no batch compaction: the reference's :256-289 shrink the batch, which changes no result).
#include <math.h>

#include "common.h"

#define CHECK_LAUNCH(ctx, what)                                                                              \
  do {                                                                                                       \
    hipError_t _e = hipGetLastError();                                                                       \
    if (_e != hipSuccess) return mhip_fail((ctx), MHIP_EHIP, what " launch: %s", hipGetErrorString(_e));    \
  } while (0)

namespace {

__global__ void beam_init_kernel(BeamState st, int* anc0, int anc_ld) {
  const int M = st.bsz * st.beam, ld = st.max_len + 2;
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < M * ld) {
    const int v = (i % ld) == 0 ? st.eos : st.pad;
    st.tokens[0][i] = v;
    st.tokens[1][i] = v;
  }
  if (i < M * anc_ld) anc0[i] = (i % anc_ld) == 0 ? i / anc_ld : 0;      // step 0: every hypothesis reads its own slot
  if (i < M) {
    st.last_tok[i] = st.eos;
    st.parent[i] = i;
    st.cum[i] = 0.f;
    st.ignore[i] = 0;
  }
  if (i < st.bsz) {
    st.finished[i] = 0;
    st.fin_count[i] = 0;
  }
  if (i == 0) *st.remaining = st.bsz;
}

__global__ void beam_select_kernel(BeamState st, int cur, int step) {
  const int s = blockIdx.x * blockDim.x + threadIdx.x;
  if (s >= st.bsz) return;
  const int beam = st.beam, K2 = 2 * beam, ld = st.max_len + 2, ML = st.max_len;
  if (st.finished[s]) {
    for (int b = 0; b < beam; ++b) { st.parent[s * beam + b] = s * beam + b; st.last_tok[s * beam + b] = st.eos; }
    return;
  }
  const float* cs = st.cand_scores + (size_t)s * K2;
  const int* ct = st.cand_tokens + (size_t)s * K2;
  const int* cb = st.cand_beams + (size_t)s * K2;
  const int* told = st.tokens[cur];
  int* tnew = st.tokens[cur ^ 1];
  bool eos_mask[8];
  for (int j = 0; j < K2; ++j) eos_mask[j] = ct[j] == st.eos && cs[j] != -INFINITY;
  for (int j = 0; j < beam; ++j)
    if (st.ignore[s * beam + j]) eos_mask[j] = false;
  // finalize_hypos: eos candidates among the first `beam`
  int nfin = st.fin_count[s];
  for (int j = 0; j < beam; ++j) {
    if (!eos_mask[j] || nfin >= beam) continue;
    const int src = s * beam + cb[j];
    int* to = st.fin_tokens + ((size_t)s * beam + nfin) * (ML + 1);
    for (int t = 0; t < step; ++t) to[t] = told[(size_t)src * ld + 1 + t];
    to[step] = st.eos;
    st.fin_len[s * beam + nfin] = step + 1;
    st.fin_score[s * beam + nfin] = cs[j] / (float)(step + 1);          // normalize_scores, len_penalty 1
    ++nfin;
  }
  st.fin_count[s] = nfin;
  if ((nfin == beam || step == ML) && (nfin > 0 || step == ML)) {
    st.finished[s] = 1;
    atomicSub(st.remaining, 1);
    for (int b = 0; b < beam; ++b) { st.parent[s * beam + b] = s * beam + b; st.last_tok[s * beam + b] = st.eos; }
    return;
  }
  // active hypotheses: the `beam` best candidates that are not finished ones
  for (int j = 0; j < beam; ++j) eos_mask[j] = eos_mask[j] || st.ignore[s * beam + j];
  int order[8], nact = 0;
  for (int j = 0; j < K2 && nact < beam; ++j)
    if (!eos_mask[j]) order[nact++] = j;
  int nign = 0;
  for (int j = 0; j < K2 && nact + nign < beam; ++j)
    if (eos_mask[j]) order[nact + nign++] = j;      // fewer than `beam` live candidates: the rest are ignored slots
  for (int b = 0; b < beam; ++b) {
    const int j = order[b], row = s * beam + b, src = s * beam + cb[j];
    st.ignore[row] = b >= nact;
    for (int t = 0; t <= step; ++t) tnew[(size_t)row * ld + t] = told[(size_t)src * ld + t];
    tnew[(size_t)row * ld + step + 1] = ct[j];
    st.parent[row] = src;
    st.last_tok[row] = ct[j];
    st.cum[row] = cs[j];
  }
}

__global__ void beam_best_kernel(BeamState st, int* tokens_out, int* lengths_out, float* scores_out) {
  const int s = blockIdx.x * blockDim.x + threadIdx.x;
  if (s >= st.bsz) return;
  const int ML = st.max_len, beam = st.beam;
  // highest score, the first finalized wins ties (torch.sort(descending) over the score list, generator.py:364-368)
  int best = -1;
  for (int j = 0; j < st.fin_count[s]; ++j)
    if (best < 0 || st.fin_score[s * beam + j] > st.fin_score[s * beam + best]) best = j;
  int* to = tokens_out + (size_t)s * (ML + 1);
  for (int t = 0; t <= ML; ++t) to[t] = st.pad;
  if (best < 0) { lengths_out[s] = 0; scores_out[s] = -INFINITY; return; }
  const int len = st.fin_len[s * beam + best];
  const int* from = st.fin_tokens + ((size_t)s * beam + best) * (ML + 1);
  for (int t = 0; t < len && t <= ML; ++t) to[t] = from[t];
  lengths_out[s] = len;
  scores_out[s] = st.fin_score[s * beam + best];
}

struct Layout {
  size_t tok0, tok1, last_tok, parent, cum, ignore, finished, fin_count, fin_tokens, fin_len, fin_score, remaining, end;
};

Layout layout(int bsz, int beam, int max_len) {
  auto up = [](size_t v) { return (v + 255) / 256 * 256; };
  const size_t M = (size_t)bsz * beam;
  Layout l;
  size_t o = 0;
  l.tok0 = o; o = up(o + M * (max_len + 2) * 4);
  l.tok1 = o; o = up(o + M * (max_len + 2) * 4);
  l.last_tok = o; o = up(o + M * 4);
  l.parent = o; o = up(o + M * 4);
  l.cum = o; o = up(o + M * 4);
  l.ignore = o; o = up(o + M);
  l.finished = o; o = up(o + bsz);
  l.fin_count = o; o = up(o + (size_t)bsz * 4);
  l.fin_tokens = o; o = up(o + M * (max_len + 1) * 4);
  l.fin_len = o; o = up(o + M * 4);
  l.fin_score = o; o = up(o + M * 4);
  l.remaining = o; o = up(o + 4);
  l.end = o;
  return l;
}

}  // namespace

size_t mhip_beam_state_bytes(int bsz, int beam, int max_len) { return layout(bsz, beam, max_len).end; }

int mhip_beam_state_carve(void* base, int bsz, int beam, int max_len, int pad, int eos, BeamState* st) {
  if (!base || !st || bsz < 1 || beam < 1 || beam > 4 || max_len < 1) return MHIP_EINVAL;
  const Layout l = layout(bsz, beam, max_len);
  char* b = (char*)base;
  st->bsz = bsz; st->beam = beam; st->max_len = max_len; st->pad = pad; st->eos = eos;
  st->tokens[0] = (int*)(b + l.tok0); st->tokens[1] = (int*)(b + l.tok1);
  st->last_tok = (int*)(b + l.last_tok); st->parent = (int*)(b + l.parent); st->cum = (float*)(b + l.cum);
  st->ignore = (unsigned char*)(b + l.ignore); st->finished = (unsigned char*)(b + l.finished);
  st->fin_count = (int*)(b + l.fin_count); st->fin_tokens = (int*)(b + l.fin_tokens); st->fin_len = (int*)(b + l.fin_len);
  st->fin_score = (float*)(b + l.fin_score); st->remaining = (int*)(b + l.remaining);
  return MHIP_OK;
}

int mhip_launch_beam_init(mhip_ctx* ctx, const BeamState& st, int* anc0, int anc_ld) {
  const long long M = (long long)st.bsz * st.beam;
  const long long total = std::max(M * (st.max_len + 2), M * anc_ld);
  PROF_LAUNCH(ctx, MHIP_K_DEC_OPS, hipLaunchKernelGGL(beam_init_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, ctx->stream, st, anc0, anc_ld));
  CHECK_LAUNCH(ctx, "beam_init");
  return 0;
}

int mhip_launch_beam_select(mhip_ctx* ctx, const BeamState& st, int cur, int step) {
  PROF_LAUNCH(ctx, MHIP_K_DEC_OPS, hipLaunchKernelGGL(beam_select_kernel, dim3((st.bsz + 63) / 64), dim3(64), 0, ctx->stream, st, cur, step));
  CHECK_LAUNCH(ctx, "beam_select");
  return 0;
}

int mhip_launch_beam_best(mhip_ctx* ctx, const BeamState& st, int* tokens_out, int* lengths_out, float* scores_out) {
  PROF_LAUNCH(ctx, MHIP_K_DEC_OPS, hipLaunchKernelGGL(beam_best_kernel, dim3((st.bsz + 63) / 64), dim3(64), 0, ctx->stream, st, tokens_out, lengths_out, scores_out));
  CHECK_LAUNCH(ctx, "beam_best");
  return 0;
}
