/*
 * Weight replication over RCCL, behind the C-ABI (no torch types, no torch bounce).
 *
 * The reference's only multi-GPU mechanism is host-staged: distribute_weights
 * (src_yolo2/network_kernels.cu:240-250) pulls every layer's weights to the host and
 * pushes them to each GPU in turn, from pthreads of ONE process.  Here the model of a
 * rank is one HBM allocation in kernel layout (the arena, y2_weights_arena), so
 * replication is ONE ncclBroadcast issued in place on that pointer, on the engine's
 * own stream: xGMI moves the 204 MB of yolo.cfg once along RCCL's ring and nothing
 * touches host memory.  One process per GPU; frames are sharded, so no other
 * collective exists on the data path.
 *
 * RCCL is bound at run time (dlopen) so that single-GPU callers of libsr_yolo2.so do
 * not need it at load time, and so that a process which already carries an RCCL (e.g.
 * the one bundled with PyTorch) uses THAT copy: a communicator must be created and
 * used by the same library instance.
 */
#define _GNU_SOURCE
#include <dlfcn.h>
#include <link.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include "y2_internal.h"

typedef struct { char internal[Y2_COMM_ID_BYTES]; } rccl_unique_id;      /* rccl.h:43 ncclUniqueId */
typedef int (*fn_get_unique_id)(rccl_unique_id *);
typedef int (*fn_comm_init_rank)(void **, int, rccl_unique_id, int);      /* the id travels by value */
typedef int (*fn_comm_destroy)(void *);
typedef int (*fn_broadcast)(const void *, void *, size_t, int, int, void *, void *);
typedef const char *(*fn_error_string)(int);
typedef int (*fn_comm_count)(void *, int *);
typedef int (*fn_all_reduce)(const void *, void *, size_t, int, int, void *, void *);

static struct {
    void *handle;
    fn_get_unique_id get_unique_id;
    fn_comm_init_rank comm_init_rank;
    fn_comm_destroy comm_destroy;
    fn_broadcast broadcast;
    fn_all_reduce all_reduce;
    fn_error_string error_string;
    fn_comm_count comm_count, comm_user_rank;
    char path[512];
} g_rccl;

static int find_loaded_rccl(struct dl_phdr_info *info, size_t size, void *data)
{
    (void)size;
    if (info->dlpi_name && strstr(info->dlpi_name, "librccl")) {
        snprintf((char *)data, sizeof g_rccl.path, "%s", info->dlpi_name);
        return 1;
    }
    return 0;
}

static int rccl_bind(void)
{
    const char *env = getenv("Y2_RCCL_LIB");
    const char *cands[] = { env, NULL, "librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1" };
    size_t i;
    if (g_rccl.handle) return 0;
    g_rccl.path[0] = 0;
    dl_iterate_phdr(find_loaded_rccl, g_rccl.path);          /* an RCCL this process already carries wins */
    cands[1] = g_rccl.path[0] ? g_rccl.path : NULL;
    for (i = 0; i < sizeof cands / sizeof cands[0] && !g_rccl.handle; ++i)
        if (cands[i] && cands[i][0]) {
            g_rccl.handle = dlopen(cands[i], RTLD_NOW | RTLD_LOCAL);
            if (g_rccl.handle && cands[i] != g_rccl.path) snprintf(g_rccl.path, sizeof g_rccl.path, "%s", cands[i]);
        }
    if (!g_rccl.handle) { y2_fail("RCCL not found (tried Y2_RCCL_LIB, a loaded librccl, librccl.so.1): %s", dlerror()); return -1; }
    g_rccl.get_unique_id = (fn_get_unique_id)dlsym(g_rccl.handle, "ncclGetUniqueId");
    g_rccl.comm_init_rank = (fn_comm_init_rank)dlsym(g_rccl.handle, "ncclCommInitRank");
    g_rccl.comm_destroy = (fn_comm_destroy)dlsym(g_rccl.handle, "ncclCommDestroy");
    g_rccl.broadcast = (fn_broadcast)dlsym(g_rccl.handle, "ncclBroadcast");
    g_rccl.all_reduce = (fn_all_reduce)dlsym(g_rccl.handle, "ncclAllReduce");
    g_rccl.error_string = (fn_error_string)dlsym(g_rccl.handle, "ncclGetErrorString");
    g_rccl.comm_count = (fn_comm_count)dlsym(g_rccl.handle, "ncclCommCount");
    g_rccl.comm_user_rank = (fn_comm_count)dlsym(g_rccl.handle, "ncclCommUserRank");
    if (!g_rccl.get_unique_id || !g_rccl.comm_init_rank || !g_rccl.comm_destroy || !g_rccl.broadcast || !g_rccl.all_reduce || !g_rccl.comm_count ||
        !g_rccl.comm_user_rank) {
        y2_fail("%s does not export the RCCL entry points", g_rccl.path);
        dlclose(g_rccl.handle);
        memset(&g_rccl, 0, sizeof g_rccl);
        return -1;
    }
    return 0;
}

static const char *rccl_err(int rc) { return g_rccl.error_string ? g_rccl.error_string(rc) : "?"; }

const char *y2_comm_library(void) { return rccl_bind() == 0 ? g_rccl.path : NULL; }

int y2_comm_unique_id(void *id_out)
{
    rccl_unique_id id;
    int rc;
    if (!id_out) { y2_fail("y2_comm_unique_id: NULL"); return -1; }
    if (rccl_bind() != 0) return -1;
    memset(&id, 0, sizeof id);
    if ((rc = g_rccl.get_unique_id(&id)) != 0) { y2_fail("ncclGetUniqueId: %s", rccl_err(rc)); return -1; }
    memcpy(id_out, &id, sizeof id);
    return 0;
}

int y2_comm_init_rank(void **comm, int nranks, const void *id_in, int rank, int device)
{
    rccl_unique_id id;
    int rc;
    if (!comm || !id_in || nranks < 1 || rank < 0 || rank >= nranks) { y2_fail("y2_comm_init_rank: bad arguments"); return -1; }
    if (rccl_bind() != 0) return -1;
    if (y2h_set_device(device) != 0) { y2_fail("y2_comm_init_rank: device %d: %s", device, y2h_last_error()); return -1; }
    memcpy(&id, id_in, sizeof id);
    *comm = NULL;
    if ((rc = g_rccl.comm_init_rank(comm, nranks, id, rank)) != 0) { y2_fail("ncclCommInitRank(rank %d of %d): %s", rank, nranks, rccl_err(rc)); return -1; }
    return 0;
}

int y2_comm_destroy(void *comm)
{
    int rc;
    if (!comm) return 0;
    if (rccl_bind() != 0) return -1;
    if ((rc = g_rccl.comm_destroy(comm)) != 0) { y2_fail("ncclCommDestroy: %s", rccl_err(rc)); return -1; }
    return 0;
}

/* ranks of the communicator and this process's rank in it, as RCCL itself reports them (ncclCommCount / ncclCommUserRank):
 * what a benchmark line quotes to show that the collective really spanned N processes */
int y2_comm_count(void *comm, int *nranks, int *rank)
{
    int rc, n = 0, r = -1;
    if (!comm) { y2_fail("y2_comm_count: NULL communicator"); return -1; }
    if (rccl_bind() != 0) return -1;
    if ((rc = g_rccl.comm_count(comm, &n)) != 0 || (rc = g_rccl.comm_user_rank(comm, &r)) != 0) {
        y2_fail("y2_comm_count: not an RCCL communicator of %s: %s", g_rccl.path, rccl_err(rc));
        return -1;
    }
    if (nranks) *nranks = n;
    if (rank) *rank = r;
    return 0;
}

/* Replicate root's packed weights to every rank of `comm`: one in-place ncclBroadcast of the arena on the engine's
 * stream.  Root must hold weights (load_weights, or a replica's resident arena); the other ranks need no load_weights at
 * all (their arena is laid out by the same plan and declared resident afterwards).  All ranks must have parsed the same
 * cfg with the same modes (strict / fp16 / fusion switches change the arena layout).  RCCL compares neither byte counts
 * nor datatypes across ranks -- a rank with another layout would hang, truncate, or fill its arena with bytes of a
 * foreign layout -- so a 16-byte handshake runs first on the same communicator: root's (arena bytes, layout signature)
 * is broadcast, every rank compares it with its own, and an all-reduce of the verdicts makes EVERY rank fail together
 * (nobody is left waiting in the big broadcast). */
int y2_broadcast_weights(network *net, void *comm, int root)
{
    y2_engine *e;
    void *arena = NULL, *d_hs = NULL;
    size_t bytes = 0;
    unsigned long long hs[4];        /* [0] bytes, [1] signature, [2..3] as int: this rank's verdict / the sum of verdicts */
    int rc, nranks = 0, rank = -1, ok, agreed = 0;
    if (!net || !comm) { y2_fail("y2_broadcast_weights: NULL network or communicator"); return -1; }
    if (y2_comm_count(comm, &nranks, &rank) != 0) return -1;
    if (root < 0 || root >= nranks) { y2_fail("y2_broadcast_weights: root %d of %d ranks", root, nranks); return -1; }
    if (rank == root) {
        if (y2_prepare(net) != 0) return -1;             /* plan + pack + upload the host weights */
    }
    if (y2_weights_arena(net, &arena, &bytes) != 0) return -1;
    e = y2_engine_of(net);
    if (!arena || !bytes || !e->stream) { y2_fail("y2_broadcast_weights: the network has no weight arena"); return -1; }
    /* handshake */
    hs[0] = (unsigned long long)bytes; hs[1] = (unsigned long long)e->arena_sig; hs[2] = hs[3] = 0;
    if (y2h_malloc(&d_hs, sizeof hs) != 0) { y2_fail("y2_broadcast_weights: %s", y2h_last_error()); return -1; }
    if (y2h_memcpy_h2d(d_hs, hs, sizeof hs, e->stream) != 0 || y2h_stream_sync(e->stream) != 0) goto hip_fail;
    if ((rc = g_rccl.broadcast(d_hs, d_hs, 16, /* ncclUint8 */ 1, root, comm, e->stream)) != 0) {
        y2_fail("ncclBroadcast of the layout handshake: %s", rccl_err(rc));
        y2h_free(d_hs);
        return -1;
    }
    {
        unsigned long long got[2] = {0, 0};
        int verdict[2];
        if (y2h_memcpy_d2h(got, d_hs, sizeof got, e->stream) != 0 || y2h_stream_sync(e->stream) != 0) goto hip_fail;
        ok = got[0] == hs[0] && got[1] == hs[1];
        verdict[0] = ok; verdict[1] = 0;
        if (y2h_memcpy_h2d((char *)d_hs + 16, verdict, sizeof verdict, e->stream) != 0 || y2h_stream_sync(e->stream) != 0) goto hip_fail;
        if ((rc = g_rccl.all_reduce((char *)d_hs + 16, (char *)d_hs + 20, 1, /* ncclInt32 */ 2, /* ncclSum */ 0, comm, e->stream)) != 0) {
            y2_fail("ncclAllReduce of the layout verdicts: %s", rccl_err(rc));
            y2h_free(d_hs);
            return -1;
        }
        if (y2h_memcpy_d2h(verdict, (char *)d_hs + 16, sizeof verdict, e->stream) != 0 || y2h_stream_sync(e->stream) != 0) goto hip_fail;
        agreed = verdict[1];
        y2h_free(d_hs); d_hs = NULL;
        if (agreed != nranks) {
            if (!ok) y2_fail("y2_broadcast_weights: rank %d laid its arena out differently from root %d (%llu bytes, signature %016llx "
                             "against %llu, %016llx): same cfg, batch-independent modes (strict / fp16 / fusion) must match on every rank",
                             rank, root, hs[0], hs[1], got[0], got[1]);
            else y2_fail("y2_broadcast_weights: %d of %d ranks have another arena layout than root %d; nothing was broadcast",
                         nranks - agreed, nranks, root);
            return -1;
        }
    }
    if ((rc = g_rccl.broadcast(arena, arena, bytes, /* ncclUint8 */ 1, root, comm, e->stream)) != 0) {
        y2_fail("ncclBroadcast of the %zu-byte weight arena: %s", bytes, rccl_err(rc));
        return -1;
    }
    if (y2h_stream_sync(e->stream) != 0) { y2_fail("y2_broadcast_weights: %s", y2h_last_error()); return -1; }
    if (rank != root) y2_weights_resident(net);
    return 0;
hip_fail:
    y2_fail("y2_broadcast_weights: %s", y2h_last_error());
    if (d_hs) y2h_free(d_hs);
    return -1;
}

/* (arena bytes, layout signature) of the current plan: what the handshake above compares; launchers that move the arena
 * by other means (torch.distributed on a view of it, MPI) exchange and compare these two numbers first */
int y2_weights_layout(network *net, unsigned long long *signature, size_t *bytes)
{
    void *arena = NULL;
    size_t b = 0;
    if (!net || y2_weights_arena(net, &arena, &b) != 0) return -1;
    if (signature) *signature = (unsigned long long)y2_engine_of(net)->arena_sig;
    if (bytes) *bytes = b;
    return 0;
}
