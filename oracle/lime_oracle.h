/*
 * lime_oracle.h -- CPU ORACLE for the LiME hot path (ClusterLCP + ClusterBWT_DA).
 *
 * THIS IS TEST INFRASTRUCTURE, NOT PRODUCT CODE.  Only tests/, __graft_entry__.smoke()
 * and bench.py's cpu_baseline leg may call it.  The product path (lime_amd/csrc, the
 * C-ABI in include/lime_hip.h) never links or loads anything from oracle/.
 *
 * It is a plain-C restatement of the reference's algorithm, written from the reference's
 * behaviour (file:line cited per function in lime_oracle.c).  Parity is PINNED: the
 * restatement is checked byte-for-byte against golden vectors produced by the reference's
 * own binaries (built from /root/reference by oracle/Makefile into oracle/_ref/) in
 * tests/test_oracle_golden.py, and live against oracle/_ref when it is present.
 */
#ifndef LIME_ORACLE_H
#define LIME_ORACLE_H
#include <stdint.h>
#include <stddef.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct { uint64_t pStart, len; } lime_oracle_cluster_t; /* == ElementCluster, Tools.h:85-88 */

/* ClusterLCP, one thread.  Returns 0, or -1 on allocation failure.  *clusters is malloc'ed
 * (free with lime_oracle_free), ascending pStart. */
int lime_oracle_detect(const uint32_t *lcp, const uint32_t *da, uint64_t n,
                       uint32_t n_reads, uint32_t alpha,
                       lime_oracle_cluster_t **clusters, uint64_t *n_clusters, uint64_t *max_len);

/* ClusterBWT_DA clusterAnalyze.  ebwt == NULL selects the EBWT=0 build.  sim is
 * n_reads*n_refs bytes row-major, accumulated into (caller zeroes).  threads<=1: serial.
 * Returns 0, -1 on alloc failure, -2 if a cluster exceeds 65536 or runs past n. */
int lime_oracle_score(const uint32_t *da, const uint8_t *ebwt, uint64_t n,
                      const lime_oracle_cluster_t *clusters, uint64_t n_clusters,
                      uint32_t n_reads, uint32_t n_refs, uint8_t *sim, int threads);

/* One (read,genome) score from two 16-bin histograms, EBWT=1 arithmetic (u8). */
uint8_t lime_oracle_pair_score(const uint8_t cr[16], const uint8_t cg[16]);

/* byte -> IUPAC index, ClusterBWT_DA.cpp:455-470 (unknown bytes -> 0). */
uint8_t lime_oracle_sym_index(uint8_t b);

/* clusterChoose row scan: per-row max and number of non-zeros. */
void lime_oracle_choose(const uint8_t *sim, uint32_t n_reads, uint32_t n_refs,
                        uint8_t *row_max, uint32_t *row_nnz);

/* clusterChoose writers.  norm = readLen+1-alpha as uint32.  Return 0 / -1 (I/O error). */
int lime_oracle_write_res_txt(const char *path, const uint8_t *sim, uint32_t n_reads,
                              uint32_t n_refs, uint32_t norm, float beta);
int lime_oracle_write_res_bin(const char *path_bin, const char *path_pos, const uint8_t *sim,
                              uint32_t n_reads, uint32_t n_refs, uint32_t norm, float beta);

/* Synthetic generator of SURVEY.md section 8(d): pure function of (seed, i). */
void lime_oracle_synth(uint64_t seed, uint64_t i0, uint64_t count, uint32_t n_reads,
                       uint32_t n_refs, uint32_t alpha, uint32_t mode,
                       uint32_t *lcp, uint32_t *da, uint8_t *ebwt);

void lime_oracle_free(void *p);

#ifdef __cplusplus
}
#endif
#endif
