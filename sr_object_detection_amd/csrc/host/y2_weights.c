/*
 * Darknet .weights container: reader and writer.
 *
 * Format (src_yolo2/parser.c:1009-1082 load_weights_upto, :963-1006
 * load_convolutional_weights, :822-878 save_weights_upto): int32 major, minor,
 * revision; `seen` as int32 when major*10+minor < 2, else uint64; then for
 * every [convolutional] layer in order: biases[n]; if batch_normalize (and not
 * dontloadscales): scales[n], rolling_mean[n], rolling_variance[n]; then
 * weights[n][c][size][size] -- raw little-endian fp32.  `flipped` layers are
 * transposed after reading (parser.c:884,997).
 *
 * The arrays read here are the HOST copies in the reference's layout; the
 * engine re-packs them into the kernel layout and uploads them at the next
 * predict (y2_engine.c upload_weights).
 */
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include "y2_internal.h"

static void transpose_in_place(float *a, int rows, int cols)
{
    float *t = calloc((size_t)rows * cols, sizeof(float));
    int r, c;
    for (r = 0; r < rows; ++r) for (c = 0; c < cols; ++c) t[(size_t)c * rows + r] = a[(size_t)r * cols + c];
    memcpy(a, t, (size_t)rows * cols * sizeof(float));
    free(t);
}

void load_weights_upto(network *net, char *filename, int cutoff)
{
    FILE *fp;
    int32_t major = 0, minor = 0, revision = 0;
    int i;
    if (!net || !net->layers) { y2_fail("load_weights: empty network"); return; }
    fprintf(stderr, "Loading weight file......");
    fp = fopen(filename, "rb");
    if (!fp) { file_error(filename); return; }
    if (fread(&major, 4, 1, fp) != 1 || fread(&minor, 4, 1, fp) != 1 || fread(&revision, 4, 1, fp) != 1) {
        fclose(fp);
        y2_fail("load_weights: %s is too short to hold a header", filename);
        return;
    }
    if (major * 10 + minor >= 2) {
        uint64_t seen = 0;
        if (fread(&seen, 8, 1, fp) != 1) seen = 0;
        memcpy(net->seen, &seen, 8);
    } else {
        int32_t seen = 0;
        if (fread(&seen, 4, 1, fp) != 1) seen = 0;
        net->seen[0] = seen;
        net->seen[1] = 0;
    }
    for (i = 0; i < net->n && i < cutoff; ++i) {
        layer *l = &net->layers[i];
        size_t num;
        if (l->dontload) continue;
        if (l->type == CONNECTED) {                  /* parser.c:897-913 load_connected_weights */
            num = (size_t)l->outputs * l->inputs;
            if (fread(l->biases, sizeof(float), l->outputs, fp) != (size_t)l->outputs) break;
            if (fread(l->weights, sizeof(float), num, fp) != num) break;
            if (major > 1000 || minor > 1000) transpose_in_place(l->weights, l->inputs, l->outputs);   /* parser.c:1035 */
            if (l->batch_normalize && !l->dontloadscales) {
                if (fread(l->scales, sizeof(float), l->outputs, fp) != (size_t)l->outputs) break;
                if (fread(l->rolling_mean, sizeof(float), l->outputs, fp) != (size_t)l->outputs) break;
                if (fread(l->rolling_variance, sizeof(float), l->outputs, fp) != (size_t)l->outputs) break;
            }
            continue;
        }
        if (l->type == BATCHNORM) {                  /* parser.c:921-931 load_batchnorm_weights */
            if (fread(l->scales, sizeof(float), l->c, fp) != (size_t)l->c) break;
            if (fread(l->rolling_mean, sizeof(float), l->c, fp) != (size_t)l->c) break;
            if (fread(l->rolling_variance, sizeof(float), l->c, fp) != (size_t)l->c) break;
            continue;
        }
        if (l->type == LOCAL) {                      /* parser.c:1068-1078 */
            num = (size_t)l->size * l->size * l->c * l->n * l->out_w * l->out_h;
            if (fread(l->biases, sizeof(float), l->outputs, fp) != (size_t)l->outputs) break;
            if (fread(l->weights, sizeof(float), num, fp) != num) break;
            continue;
        }
        if (l->type != CONVOLUTIONAL) continue;
        num = (size_t)l->n * l->c * l->size * l->size;
        /* short reads leave the remaining values untouched, as in the reference */
        if (fread(l->biases, sizeof(float), l->n, fp) != (size_t)l->n) break;
        if (l->batch_normalize && !l->dontloadscales) {
            if (fread(l->scales, sizeof(float), l->n, fp) != (size_t)l->n) break;
            if (fread(l->rolling_mean, sizeof(float), l->n, fp) != (size_t)l->n) break;
            if (fread(l->rolling_variance, sizeof(float), l->n, fp) != (size_t)l->n) break;
        }
        if (fread(l->weights, sizeof(float), num, fp) != num) break;
        if (l->flipped) transpose_in_place(l->weights, l->c * l->size * l->size, l->n);
    }
    fclose(fp);
    fprintf(stderr, "Done!\n");
    if (y2_engine_of(net)) {
        y2_engine_of(net)->weights_dirty = 1;
        y2_engine_of(net)->weights_external = 0;
    }
}

void load_weights(network *net, char *filename) { load_weights_upto(net, filename, net ? net->n : 0); }

void save_weights_upto(network net, char *filename, int cutoff)
{
    FILE *fp = fopen(filename, "wb");
    int32_t hdr[4];
    int i;
    fprintf(stderr, "Saving weights to %s\n", filename);
    if (!fp) { file_error(filename); return; }
    hdr[0] = 0; hdr[1] = 1; hdr[2] = 0;       /* version 0.1.0: 32-bit seen (parser.c:833-839) */
    hdr[3] = net.seen ? net.seen[0] : 0;
    fwrite(hdr, 4, 4, fp);
    for (i = 0; i < net.n && i < cutoff; ++i) {
        layer *l = &net.layers[i];
        if (l->type == CONNECTED) {                  /* parser.c:806-820 save_connected_weights */
            fwrite(l->biases, sizeof(float), l->outputs, fp);
            fwrite(l->weights, sizeof(float), (size_t)l->outputs * l->inputs, fp);
            if (l->batch_normalize) {
                fwrite(l->scales, sizeof(float), l->outputs, fp);
                fwrite(l->rolling_mean, sizeof(float), l->outputs, fp);
                fwrite(l->rolling_variance, sizeof(float), l->outputs, fp);
            }
            continue;
        }
        if (l->type == BATCHNORM) {                  /* parser.c:794-804 save_batchnorm_weights */
            fwrite(l->scales, sizeof(float), l->c, fp);
            fwrite(l->rolling_mean, sizeof(float), l->c, fp);
            fwrite(l->rolling_variance, sizeof(float), l->c, fp);
            continue;
        }
        if (l->type == LOCAL) {                      /* parser.c:865-875 */
            fwrite(l->biases, sizeof(float), l->outputs, fp);
            fwrite(l->weights, sizeof(float), (size_t)l->size * l->size * l->c * l->n * l->out_w * l->out_h, fp);
            continue;
        }
        if (l->type != CONVOLUTIONAL) continue;
        fwrite(l->biases, sizeof(float), l->n, fp);
        if (l->batch_normalize) {
            fwrite(l->scales, sizeof(float), l->n, fp);
            fwrite(l->rolling_mean, sizeof(float), l->n, fp);
            fwrite(l->rolling_variance, sizeof(float), l->n, fp);
        }
        fwrite(l->weights, sizeof(float), (size_t)l->n * l->c * l->size * l->size, fp);
    }
    fclose(fp);
}

/* convolutional_layer.c:321-334: fold the batch-norm statistics into weights and biases (weight-file surgery for
 * `darknet denormalize`, darknet.c:309).  Note the reference's own constant here, sqrt(var + .00001), differs from
 * the forward pass's sqrt(var) + .000001f.  The caller clears batch_normalize afterwards, as denormalize_net does. */
void denormalize_convolutional_layer(layer l)
{
    int i, j;
    const int per = l.c * l.size * l.size;
    if (l.type != CONVOLUTIONAL || !l.weights || !l.scales) return;
    for (i = 0; i < l.n; ++i) {
        float scale = l.scales[i] / sqrt(l.rolling_variance[i] + .00001);
        for (j = 0; j < per; ++j) l.weights[(size_t)i * per + j] *= scale;
        l.biases[i] -= l.rolling_mean[i] * scale;
        l.scales[i] = 1;
        l.rolling_mean[i] = 0;
        l.rolling_variance[i] = 1;
    }
    if (l.dev && ((y2_ldev *)l.dev)->eng) ((y2_ldev *)l.dev)->eng->weights_dirty = 1;   /* re-pack at the next forward */
}

/* darknet.c:309-345 denormalize_net for the layer types this engine has */
void y2_denormalize_network(network *net)
{
    int i;
    for (i = 0; i < net->n; ++i) {
        layer l = net->layers[i];
        if (l.type == CONVOLUTIONAL && l.batch_normalize) {
            denormalize_convolutional_layer(l);
            net->layers[i].batch_normalize = 0;
        }
    }
    y2_engine_invalidate(net);      /* the arena layout depends on batch_normalize */
}

void save_weights(network net, char *filename) { save_weights_upto(net, filename, net.n); }
