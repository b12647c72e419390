/* Plain-C client of include/hannoy_amd.h: builds the KAT-1 index
 * (/root/reference/src/tests/writer.rs:376-408) through hny_build and prints every record the
 * reference's write loop + Writer::build tail would put, as "key_hex value_hex" lines.
 * Exit code 0 = built, 2 = no GPU (HNY_ERR_NO_DEVICE), 1 = anything else. */
#include <stdio.h>
#include <string.h>

#include "hannoy_amd.h"

static int sink(void *ctx, const uint8_t *key, size_t kl, const uint8_t *val, size_t vl) {
  (void)ctx;
  for (size_t i = 0; i < kl; i++) printf("%02x", key[i]);
  printf(" ");
  for (size_t i = 0; i < vl; i++) printf("%02x", val[i]);
  printf("\n");
  return 0;
}

/* this translation unit's view of the public structs (gcc, C99) against the library's own (hipcc, C++) */
static int abi_matches(void) {
  uint32_t s[HNY_ABI_N_STRUCTS];
  if (hny_abi_sizes(s, HNY_ABI_N_STRUCTS) != HNY_ABI_N_STRUCTS) return 0;
  return s[HNY_ABI_BUILD_OPTS] == sizeof(hny_build_opts) && s[HNY_ABI_ITEMS] == sizeof(hny_items) &&
         s[HNY_ABI_GRAPH] == sizeof(hny_graph) && s[HNY_ABI_PREV_GRAPH] == sizeof(hny_prev_graph) &&
         s[HNY_ABI_BATCH] == sizeof(hny_batch) && s[HNY_ABI_QUERY_OPTS] == sizeof(hny_query_opts) &&
         s[HNY_ABI_LMDB_STAT] == sizeof(hny_lmdb_stat);
}

int main(void) {
  if (!abi_matches()) {
    fprintf(stderr, "ABI mismatch: struct sizes differ between hannoy_amd.h as compiled here and the library\n");
    return 3;
  }
  float vecs[6][2];
  uint32_t ids[6];
  uint8_t levels[6] = {1, 0, 1, 1, 0, 0};
  for (int i = 0; i < 6; i++) {
    vecs[i][0] = (float)i;
    vecs[i][1] = 0.0f;
    ids[i] = (uint32_t)i;
  }
  uint8_t codes[6 * 8], headers[6 * 4];
  if (hny_vector_bytes(HNY_EUCLIDEAN, 2) != 8 || hny_header_bytes(HNY_EUCLIDEAN) != 4) return 1;
  if (hny_encode_vectors(HNY_EUCLIDEAN, 2, 6, &vecs[0][0], codes, headers) != HNY_OK) return 1;

  hny_build_opts o;
  memset(&o, 0, sizeof o);
  o.metric = HNY_EUCLIDEAN;
  o.dim = 2;
  o.M = 3;
  o.M0 = 3;
  o.ef_construction = 100;
  o.alpha = 1.0f;
  o.batch_max = 1; /* strictly sequential insertion, like the reference's test (1 rayon thread) */
  o.device = -1;
  hny_items it;
  memset(&it, 0, sizeof it);
  it.n = 6;
  it.ids = ids;
  it.vectors = codes;
  it.stride = 8;
  it.headers = headers;
  it.header_size = 4;
  it.levels = levels;

  hny_graph *g = NULL;
  int rc = hny_build(&o, &it, &g);
  if (rc == HNY_ERR_NO_DEVICE) {
    fprintf(stderr, "no device: %s\n", hny_last_error());
    return 2;
  }
  if (rc != HNY_OK) {
    fprintf(stderr, "hny_build failed (%d): %s\n", rc, hny_last_error());
    return 1;
  }
  fprintf(stderr, "%s: %llu records, %u entry points, max_level %u\n", hny_version(),
          (unsigned long long)g->n_records, g->n_entry_points, g->max_level);
  rc = hny_encode_kv(g, &o, &it, 0, 1, sink, NULL);
  hny_graph_free(g);
  return rc == HNY_OK ? 0 : 1;
}
