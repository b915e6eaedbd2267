/*
 * Pinned, multi-buffered host feed: frames that start in HOST memory reach the network without stalling it.
 *
 * The reference pays, per call, cudaMalloc + a pageable H2D copy + cudaFree in front of every forward
 * (src_yolo2/network_kernels.cu:392-405 network_predict_gpu), all on the compute stream.  Here a network owns
 * `slots` pairs of (pinned host buffer, HBM buffer) and a copy stream:
 *
 *     producer fills y2_feed_host(net, s)        (camera / decoder writes straight into pinned memory)
 *     y2_feed_submit(net, s, bytes)              H2D on the copy stream, asynchronous
 *     y2_feed_forward(net, s)                    engine stream waits (on the device) for that copy, then runs the forward
 *
 * so the upload of batch i+1 rides the PCIe link while batch i computes; with two slots the link (about 50 GB/s) carries
 * a 141 MB fp32 batch of yolo.cfg 608x608 b32 in under 3 ms against a 15 ms forward, and a u8 camera batch in a
 * quarter of that.  Ordering is by events only, the host never blocks except in y2_feed_wait_host (before it
 * overwrites a pinned buffer whose copy may still be in flight).
 */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include "y2_internal.h"

#define HIPCALL_I(expr) do { int rc_ = (expr); if (rc_ != 0) { y2_fail("%s failed (%d): %s", #expr, rc_, y2h_last_error()); return -1; } } while (0)

void y2_feed_close(network *net)
{
    y2_engine *e = y2_engine_of(net);
    int i;
    if (!e || !e->feed_slots) return;
    y2h_set_device(e->device);
    if (e->feed_stream) y2h_stream_sync(e->feed_stream);
    for (i = 0; i < e->feed_slots; ++i) {
        if (e->feed_host) y2h_host_free(e->feed_host[i]);
        if (e->feed_dev) y2h_free(e->feed_dev[i]);
        if (e->feed_up) y2h_event_destroy(e->feed_up[i]);
        if (e->feed_done) y2h_event_destroy(e->feed_done[i]);
    }
    free(e->feed_host); free(e->feed_dev); free(e->feed_up); free(e->feed_done); free(e->feed_used);
    e->feed_host = e->feed_dev = NULL; e->feed_up = e->feed_done = NULL; e->feed_used = NULL;
    y2h_stream_destroy(e->feed_stream); e->feed_stream = NULL;
    e->feed_slots = 0; e->feed_bytes = 0;
}

/* slot_bytes 0: one batch of float NCHW frames (batch * inputs * 4) */
int y2_feed_open(network *net, int slots, size_t slot_bytes)
{
    y2_engine *e;
    int i;
    if (!net || slots < 1 || slots > 16) { y2_fail("y2_feed_open: 1..16 slots"); return -1; }
    if (y2_prepare(net) != 0) return -1;
    e = y2_engine_of(net);
    y2_feed_close(net);
    HIPCALL_I(y2h_set_device(e->device));
    if (!slot_bytes) slot_bytes = e->in_floats * sizeof(float);
    e->feed_host = calloc(slots, sizeof(void *)); e->feed_dev = calloc(slots, sizeof(void *));
    e->feed_up = calloc(slots, sizeof(y2h_event)); e->feed_done = calloc(slots, sizeof(y2h_event));
    e->feed_used = calloc(slots, sizeof(int));
    if (!e->feed_host || !e->feed_dev || !e->feed_up || !e->feed_done || !e->feed_used) {
        free(e->feed_host); free(e->feed_dev); free(e->feed_up); free(e->feed_done); free(e->feed_used);
        e->feed_host = e->feed_dev = NULL; e->feed_up = e->feed_done = NULL; e->feed_used = NULL;
        y2_fail("y2_feed_open: out of memory");
        return -1;
    }
    e->feed_slots = slots; e->feed_bytes = slot_bytes;
    HIPCALL_I(y2h_stream_create(&e->feed_stream));
    for (i = 0; i < slots; ++i) {
        HIPCALL_I(y2h_host_alloc(&e->feed_host[i], slot_bytes));
        HIPCALL_I(y2h_malloc(&e->feed_dev[i], slot_bytes));
        HIPCALL_I(y2h_event_create(&e->feed_up[i]));
        HIPCALL_I(y2h_event_create(&e->feed_done[i]));
    }
    return 0;
}

static y2_engine *feed_of(network *net, int slot, const char *who)
{
    y2_engine *e = y2_engine_of(net);
    if (!e || !e->feed_slots) { y2_fail("%s: call y2_feed_open first", who); return NULL; }
    if (slot < 0 || slot >= e->feed_slots) { y2_fail("%s: slot %d of %d", who, slot, e->feed_slots); return NULL; }
    return e;
}

void *y2_feed_host(network net, int slot)
{
    y2_engine *e = feed_of(&net, slot, "y2_feed_host");
    return e ? e->feed_host[slot] : NULL;
}

void *y2_feed_device(network net, int slot)
{
    y2_engine *e = feed_of(&net, slot, "y2_feed_device");
    return e ? e->feed_dev[slot] : NULL;
}

size_t y2_feed_slot_bytes(network net) { y2_engine *e = y2_engine_of(&net); return e ? e->feed_bytes : 0; }

int y2_feed_submit(network net, int slot, size_t bytes)
{
    y2_engine *e = feed_of(&net, slot, "y2_feed_submit");
    if (!e) return -1;
    if (!bytes) bytes = e->feed_bytes;
    if (bytes > e->feed_bytes) { y2_fail("y2_feed_submit: %zu bytes into a %zu-byte slot", bytes, e->feed_bytes); return -1; }
    HIPCALL_I(y2h_set_device(e->device));
    /* the HBM copy of this slot may still be read by the forward that was fed from it */
    if (e->feed_used[slot]) HIPCALL_I(y2h_stream_wait_event(e->feed_stream, e->feed_done[slot]));
    HIPCALL_I(y2h_memcpy_h2d(e->feed_dev[slot], e->feed_host[slot], bytes, e->feed_stream));
    HIPCALL_I(y2h_event_record(e->feed_up[slot], e->feed_stream));
    return 0;
}

int y2_feed_wait_host(network net, int slot)
{
    y2_engine *e = feed_of(&net, slot, "y2_feed_wait_host");
    if (!e) return -1;
    HIPCALL_I(y2h_event_sync(e->feed_up[slot]));
    return 0;
}

int y2_feed_forward(network net, int slot)
{
    y2_engine *e = feed_of(&net, slot, "y2_feed_forward");
    if (!e) return -1;
    if (e->feed_bytes < e->in_floats * sizeof(float)) { y2_fail("y2_feed_forward: the slots hold %zu bytes, a float batch needs %zu", e->feed_bytes, e->in_floats * sizeof(float)); return -1; }
    HIPCALL_I(y2h_set_device(e->device));
    HIPCALL_I(y2h_stream_wait_event(e->stream, e->feed_up[slot]));
    if (y2_forward_device(net, (const float *)e->feed_dev[slot]) != 0) return -1;
    HIPCALL_I(y2h_event_record(e->feed_done[slot], e->stream));
    e->feed_used[slot] = 1;
    return 0;
}

int y2_feed_forward_u8(network net, int slot, int h, int w, int c, int step, int swap_rb, int letterbox)
{
    y2_engine *e = feed_of(&net, slot, "y2_feed_forward_u8");
    if (!e) return -1;
    if (h <= 0 || step <= 0 || (size_t)step * h * net.batch > e->feed_bytes) { y2_fail("y2_feed_forward_u8: %d frames of %d x %d bytes do not fit a %zu-byte slot", net.batch, h, step, e->feed_bytes); return -1; }
    HIPCALL_I(y2h_set_device(e->device));
    HIPCALL_I(y2h_stream_wait_event(e->stream, e->feed_up[slot]));
    if (y2_ingest_u8_device(net, (const unsigned char *)e->feed_dev[slot], h, w, c, step, swap_rb, letterbox) != 0) return -1;
    /* the conversion kernels have read the slot once they are done: the forward itself reads the network's own input buffer */
    HIPCALL_I(y2h_event_record(e->feed_done[slot], e->stream));
    e->feed_used[slot] = 1;
    return y2_forward_device(net, NULL);
}
