/* A plain C caller of include/opty_hip.h (test infrastructure): what a
 * non-Python host -- or the reference's maintainer writing the binding by
 * hand -- does with the library.  No Python, no torch, no C++:
 *
 *   abi_client <case file> <output file> [shard]
 *
 * The case file is written by tests/test_c_client.py: the opty_hip_desc of a
 * problem as raw bytes, the path of its gfx950 code object, the known
 * parameters, the node time interval, the instance-constraint index tables and
 * a free vector.  The client creates a handle, evaluates constraints(free),
 * jacobian(free) and jacobian_indices() into malloc'ed HOST buffers and writes
 * them out; the test compares them with the Python host side's. */
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "opty_hip.h"

#define DIE(...) do { fprintf(stderr, __VA_ARGS__); fprintf(stderr, "\n"); \
                      return 2; } while (0)
#define TRY(call) do { if ((call) != 0) DIE("%s: %s", #call, \
                                           opty_hip_last_error()); } while (0)

static int read_exact(FILE *f, void *dst, size_t bytes) {
    return bytes == 0 || fread(dst, 1, bytes, f) == bytes;
}

/* The node-sharded path from plain C, in a world of one rank: bring-up of the
 * library's own RCCL communicator, the global free vector "broadcast", the
 * node range evaluated into shard buffers (a dense (M, nodes) constraint block
 * and the Jacobian slice) from device memory, gather-v into the global
 * vectors (strided constraint copy, Jacobian in place), instance tails by the
 * root.  Same outputs as the host-buffer calls, bit for bit. */
static int sharded(opty_hip_problem *h, const opty_hip_desc *desc,
                   const double *free_vec, int64_t num_free, double *con,
                   double *jac) {
    const int64_t ncn = desc->N - 1, ncon = opty_hip_num_constraints(h),
                  nnz = opty_hip_nnz(h);
    unsigned char id[OPTY_HIP_COMM_ID_BYTES];
    opty_hip_comm *comm = NULL;
    TRY(opty_hip_comm_unique_id(id));
    TRY(opty_hip_comm_create(id, 0, 1, desc->device, &comm));
    if (opty_hip_comm_rank(comm) != 0 || opty_hip_comm_world(comm) != 1)
        DIE("rank / world of a world of one");
    double *d_free = opty_hip_device_alloc(desc->device,
                                           sizeof(double)*(size_t)num_free);
    double *d_con = opty_hip_device_alloc(desc->device,
                                          sizeof(double)*(size_t)ncon);
    double *d_jac = opty_hip_device_alloc(desc->device,
                                          sizeof(double)*(size_t)nnz);
    double *s_con = opty_hip_device_alloc(
        desc->device, sizeof(double)*(size_t)(desc->M*ncn));
    double *s_jac = opty_hip_device_alloc(
        desc->device, sizeof(double)*(size_t)(desc->P*ncn));
    if (!d_free || !d_con || !d_jac || !s_con || !s_jac)
        DIE("device_alloc: %s", opty_hip_last_error());
    TRY(opty_hip_memcpy(d_free, free_vec, sizeof(double)*(size_t)num_free, 0));
    TRY(opty_hip_bcast_free(comm, h, d_free, 0));
    /* the rank's node range [0, ncn) in two launches, into shard buffers */
    const int64_t mid = ncn/2, bounds[2] = {0, ncn};
    TRY(opty_hip_eval_shard(h, OPTY_HIP_EVAL_FUSED, d_free, s_con, ncn, s_jac,
                            0, mid));
    TRY(opty_hip_eval_shard(h, OPTY_HIP_EVAL_PAIR, d_free, s_con + mid, ncn,
                            s_jac + mid*desc->P, mid, ncn));
    TRY(opty_hip_gather_v(comm, h, bounds, s_con, s_jac, d_con, d_jac, 0,
                          OPTY_HIP_EVAL_PAIR));
    if (desc->num_inst > 0)
        TRY(opty_hip_eval_instance(h, d_free, d_con + desc->M*ncn,
                                   d_jac + desc->P*ncn));
    TRY(opty_hip_synchronize(h));
    TRY(opty_hip_memcpy(con, d_con, sizeof(double)*(size_t)ncon, 1));
    TRY(opty_hip_memcpy(jac, d_jac, sizeof(double)*(size_t)nnz, 1));
    /* misuse is reported */
    if (opty_hip_gather_v(comm, h, bounds, s_con, s_jac, NULL, d_jac, 0,
                          OPTY_HIP_EVAL_PAIR) == 0)
        DIE("a root without its global vector was accepted");
    if (opty_hip_bcast_free(comm, h, d_free, 3) == 0)
        DIE("a root outside the world was accepted");
    TRY(opty_hip_comm_destroy(comm));
    TRY(opty_hip_device_free(d_free));
    TRY(opty_hip_device_free(d_con));
    TRY(opty_hip_device_free(d_jac));
    TRY(opty_hip_device_free(s_con));
    TRY(opty_hip_device_free(s_jac));
    return 0;
}

int main(int argc, char **argv) {
    if (argc != 3 && !(argc == 4 && strcmp(argv[3], "shard") == 0))
        DIE("usage: abi_client <case> <out> [shard]");
    if (opty_hip_abi_version() != OPTY_HIP_ABI_VERSION)
        DIE("libopty_hip.so implements ABI %d, this client was built for %d",
            opty_hip_abi_version(), OPTY_HIP_ABI_VERSION);
    FILE *f = fopen(argv[1], "rb");
    if (!f) DIE("cannot open %s", argv[1]);
    opty_hip_desc desc;
    int64_t path_len = 0, num_free = 0;
    double interval = 0.0;
    if (!read_exact(f, &desc, sizeof desc) ||
        !read_exact(f, &path_len, sizeof path_len))
        DIE("short case file");
    char *path = calloc((size_t)path_len + 1, 1);
    double *params = malloc(sizeof(double)*(size_t)(desc.p_known + 1));
    int64_t *atoms = malloc(sizeof(int64_t)*(size_t)(desc.num_inst_atoms + 1));
    int64_t *irows = malloc(sizeof(int64_t)*(size_t)(desc.nnz_inst + 1));
    int64_t *icols = malloc(sizeof(int64_t)*(size_t)(desc.nnz_inst + 1));
    if (!read_exact(f, path, (size_t)path_len) ||
        !read_exact(f, params, sizeof(double)*(size_t)desc.p_known) ||
        !read_exact(f, &interval, sizeof interval) ||
        !read_exact(f, atoms, sizeof(int64_t)*(size_t)desc.num_inst_atoms) ||
        !read_exact(f, irows, sizeof(int64_t)*(size_t)desc.nnz_inst) ||
        !read_exact(f, icols, sizeof(int64_t)*(size_t)desc.nnz_inst) ||
        !read_exact(f, &num_free, sizeof num_free))
        DIE("short case file");
    double *free_vec = malloc(sizeof(double)*(size_t)num_free);
    if (!read_exact(f, free_vec, sizeof(double)*(size_t)num_free))
        DIE("short case file");
    fclose(f);

    if (opty_hip_device_count() < 1) DIE("no HIP device");
    opty_hip_problem *h = NULL;
    TRY(opty_hip_create(&desc, path, &h));
    if (opty_hip_num_free(h) != num_free) DIE("num_free mismatch");
    if (desc.p_known > 0)
        TRY(opty_hip_set_known_parameters(h, params, desc.p_known));
    if (desc.s == 0) TRY(opty_hip_set_interval(h, interval));
    if (desc.num_inst > 0)
        TRY(opty_hip_set_instance_indices(h, atoms, irows, icols));

    const int64_t ncon = opty_hip_num_constraints(h), nnz = opty_hip_nnz(h);
    double *con = malloc(sizeof(double)*(size_t)ncon);
    double *jac = malloc(sizeof(double)*(size_t)nnz);
    int64_t *rows = malloc(sizeof(int64_t)*(size_t)nnz);
    int64_t *cols = malloc(sizeof(int64_t)*(size_t)nnz);
    if (argc == 4) {
        int rc = sharded(h, &desc, free_vec, num_free, con, jac);
        if (rc != 0) return rc;
    } else {
        TRY(opty_hip_eval_con(h, free_vec, con, OPTY_HIP_HOST));
        TRY(opty_hip_eval_jac(h, free_vec, jac, OPTY_HIP_HOST));
    }
    TRY(opty_hip_jacobian_indices(h, rows, cols, OPTY_HIP_HOST));
    /* misuse is reported, not fatal */
    if (opty_hip_eval_con(h, NULL, con, OPTY_HIP_HOST) == 0)
        DIE("a null free vector was accepted");
    if (strstr(opty_hip_last_error(), "null") == NULL)
        DIE("unexpected message: %s", opty_hip_last_error());
    TRY(opty_hip_destroy(h));

    FILE *o = fopen(argv[2], "wb");
    if (!o) DIE("cannot write %s", argv[2]);
    fwrite(&ncon, sizeof ncon, 1, o);
    fwrite(&nnz, sizeof nnz, 1, o);
    fwrite(con, sizeof(double), (size_t)ncon, o);
    fwrite(jac, sizeof(double), (size_t)nnz, o);
    fwrite(rows, sizeof(int64_t), (size_t)nnz, o);
    fwrite(cols, sizeof(int64_t), (size_t)nnz, o);
    fclose(o);
    printf("abi_client: %lld constraints, %lld Jacobian values\n",
           (long long)ncon, (long long)nnz);
    return 0;
}
