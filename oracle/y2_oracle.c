/*
 * y2_oracle.c -- CPU ORACLE.  TEST INFRASTRUCTURE ONLY.
 *
 * A from-scratch restatement, in plain C, of the reference's CPU forward path
 * (YOLOv2 / Darknet-19: cfg parse -> .weights load -> network_predict ->
 * get_region_boxes -> do_nms_sort -> Detector / test_detector_img hand-off).
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may
 * load this file's shared object; the product library never links it.
 *
 * Parity status: PINNED.  The reference holds no golden vectors for this path
 * (SURVEY.md section 4), so this oracle is pinned against outputs of the
 * reference itself compiled here (oracle/build_ref.sh -> oracle/_ref/) --
 * tests/test_oracle_vs_reference.py requires bit-identical tensors when
 * /root/reference is present, and tests/golden/ holds vectors produced by the
 * compiled reference (tests/golden/gen_golden.py) for when it is not.
 *
 * Every function cites the reference lines (relative to src_yolo2/) whose
 * operation order and rounding it follows.  Build with
 *   gcc -O2 -fopenmp -ffp-contract=off  (no -march, no -ffast-math)
 * so that every float multiply/add rounds separately, as in the reference's
 * x86-64 baseline build.
 */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <math.h>
#include <float.h>
#include <stdint.h>
#include <time.h>

enum { ORC_CONV = 0, ORC_MAXPOOL, ORC_ROUTE, ORC_REORG, ORC_REGION, ORC_AVGPOOL, ORC_SOFTMAX, ORC_COST, ORC_SHORTCUT,
       ORC_CONNECTED, ORC_DROPOUT, ORC_DETECTION, ORC_CROP, ORC_LOCAL, ORC_BATCHNORM };
enum { ACT_LOGISTIC = 0, ACT_RELU, ACT_LINEAR, ACT_LEAKY, ACT_RAMP, ACT_TANH, ACT_ELU, ACT_HARDTAN,
       ACT_RELIE, ACT_PLSE, ACT_LOGGY, ACT_STAIR, ACT_LHTAN };

typedef struct {
    int n;            /* nodes */
    int *parent;
    int *group;
    int *leaf;
    int groups;
    int *group_size;
    int *group_offset;
} orc_tree;

typedef struct {
    int kind;
    int batch;
    int w, h, c;
    int out_w, out_h, out_c;
    int inputs, outputs;
    int n;             /* filters | anchors | number of route inputs */
    int size, stride, pad;
    int batch_normalize, activation, flipped;
    int reverse;
    int classes, coords, softmax, classfix;
    int side, sqrt_;   /* [detection] */
    int noadjust;      /* [crop] */
    int xnor;          /* [convolutional] xnor=1 */
    int groups;
    float temperature;
    int dontload, dontloadscales;
    float *weights, *biases, *scales, *rolling_mean, *rolling_variance;
    float *output;
    int *input_layers, *input_sizes;
    orc_tree *tree;
    int *map;
    int map_n;
} orc_layer;

typedef struct {
    int n, batch, subdivisions;
    int w, h, c, inputs;
    uint64_t seen;
    orc_layer *layers;
    float *workspace;
    size_t workspace_floats;
} orc_net;

/* ------------------------------------------------------------------ */
/* cfg text -> sections  (parser.c:702-735 read_cfg, utils.c:230 strip, */
/* utils.c:263 fgetl, option_list.c:35 read_option, :74 option_find)    */
/* ------------------------------------------------------------------ */
typedef struct kv { char *key, *val; struct kv *next; } kv;
typedef struct section { char *type; kv *head, *tail; struct section *next; } section;

static char *read_line(FILE *fp)
{
    size_t cap = 256, len = 0;
    char *s = malloc(cap);
    int ch;
    if (feof(fp)) { free(s); return NULL; }
    while ((ch = fgetc(fp)) != EOF && ch != '\n') {
        if (len + 2 > cap) { cap *= 2; s = realloc(s, cap); }
        s[len++] = (char)ch;
    }
    if (ch == EOF && len == 0) { free(s); return NULL; }
    s[len] = 0;
    return s;
}

static void strip_ws(char *s)   /* every blank, tab, CR, LF is removed, not only the ends */
{
    char *d = s;
    for (; *s; ++s) if (*s != ' ' && *s != '\t' && *s != '\n' && *s != '\r') *d++ = *s;
    *d = 0;
}

static section *read_sections(const char *path, int *count)
{
    FILE *fp = fopen(path, "r");
    section *first = NULL, *cur = NULL;
    char *line;
    int nsec = 0, lineno = 0;
    if (!fp) { fprintf(stderr, "oracle: cannot open cfg %s\n", path); return NULL; }
    while ((line = read_line(fp)) != NULL) {
        ++lineno;
        strip_ws(line);
        if (line[0] == '[') {
            section *s = calloc(1, sizeof *s);
            s->type = line;
            if (cur) cur->next = s; else first = s;
            cur = s;
            ++nsec;
        } else if (line[0] == 0 || line[0] == '#' || line[0] == ';') {
            free(line);
        } else {
            char *eq = strchr(line, '=');
            size_t len = strlen(line);
            if ((eq && (size_t)(eq - line) == len - 1) || !cur) {   /* "key=" with empty value: rejected */
                fprintf(stderr, "Config file error line %d, could parse: %s\n", lineno, line);
                free(line);
                continue;
            }
            kv *p = calloc(1, sizeof *p);
            p->key = line;
            if (eq) { *eq = 0; p->val = eq + 1; }
            if (cur->tail) cur->tail->next = p; else cur->head = p;
            cur->tail = p;
        }
    }
    fclose(fp);
    if (count) *count = nsec;
    return first;
}

static const char *opt(section *s, const char *key)   /* first matching key wins */
{
    kv *p;
    for (p = s->head; p; p = p->next) if (strcmp(p->key, key) == 0) return p->val;
    return NULL;
}
static int opt_int(section *s, const char *key, int def) { const char *v = opt(s, key); return v ? atoi(v) : def; }
static float opt_float(section *s, const char *key, float def) { const char *v = opt(s, key); return v ? (float)atof(v) : def; }

static void free_sections(section *s)
{
    while (s) {
        section *n = s->next;
        kv *p = s->head;
        while (p) { kv *q = p->next; free(p->key); free(p); p = q; }
        free(s->type);
        free(s);
        s = n;
    }
}

static int activation_from_name(const char *s)   /* activations.c get_activation */
{
    if (!strcmp(s, "logistic")) return ACT_LOGISTIC;
    if (!strcmp(s, "relu")) return ACT_RELU;
    if (!strcmp(s, "linear")) return ACT_LINEAR;
    if (!strcmp(s, "leaky")) return ACT_LEAKY;
    if (!strcmp(s, "ramp")) return ACT_RAMP;
    if (!strcmp(s, "tanh")) return ACT_TANH;
    if (!strcmp(s, "elu")) return ACT_ELU;
    if (!strcmp(s, "hardtan")) return ACT_HARDTAN;
    if (!strcmp(s, "relie")) return ACT_RELIE;
    if (!strcmp(s, "plse")) return ACT_PLSE;
    if (!strcmp(s, "loggy")) return ACT_LOGGY;
    if (!strcmp(s, "stair")) return ACT_STAIR;
    if (!strcmp(s, "lhtan")) return ACT_LHTAN;
    fprintf(stderr, "Couldn't find activation function %s, going with ReLU\n", s);
    return ACT_RELU;
}

/* tree.c:53 read_tree: one "name parent" line per node; a new group starts whenever the parent changes */
orc_tree *orc_read_tree(const char *path)
{
    FILE *fp = fopen(path, "r");
    orc_tree *t;
    char *line;
    int last_parent = -1, group_size = 0, groups = 0, n = 0;
    if (!fp) { fprintf(stderr, "oracle: cannot open tree %s\n", path); return NULL; }
    t = calloc(1, sizeof *t);
    while ((line = read_line(fp)) != NULL) {
        char id[512];
        int parent = -1;
        sscanf(line, "%511s %d", id, &parent);
        free(line);
        t->parent = realloc(t->parent, (n + 1) * sizeof(int));
        t->parent[n] = parent;
        if (parent != last_parent) {
            ++groups;
            t->group_offset = realloc(t->group_offset, groups * sizeof(int));
            t->group_size = realloc(t->group_size, groups * sizeof(int));
            t->group_offset[groups - 1] = n - group_size;
            t->group_size[groups - 1] = group_size;
            group_size = 0;
            last_parent = parent;
        }
        t->group = realloc(t->group, (n + 1) * sizeof(int));
        t->group[n] = groups;
        ++n;
        ++group_size;
    }
    ++groups;
    t->group_offset = realloc(t->group_offset, groups * sizeof(int));
    t->group_size = realloc(t->group_size, groups * sizeof(int));
    t->group_offset[groups - 1] = n - group_size;
    t->group_size[groups - 1] = group_size;
    t->n = n;
    t->groups = groups;
    t->leaf = calloc(n, sizeof(int));
    { int i; for (i = 0; i < n; ++i) t->leaf[i] = 1; for (i = 0; i < n; ++i) if (t->parent[i] >= 0) t->leaf[t->parent[i]] = 0; }
    fclose(fp);
    return t;
}

static int *read_int_lines(const char *path, int *n_out)   /* utils.c:17 read_map */
{
    FILE *fp = fopen(path, "r");
    int n = 0, *m = NULL;
    char *line;
    if (!fp) { fprintf(stderr, "oracle: cannot open map %s\n", path); return NULL; }
    while ((line = read_line(fp)) != NULL) {
        m = realloc(m, (n + 1) * sizeof(int));
        m[n++] = atoi(line);
        free(line);
    }
    fclose(fp);
    *n_out = n;
    return m;
}

/* ------------------------------------------------------------------ */
/* parse_network_cfg  (parser.c:585-700) and the per-layer make_*      */
/* ------------------------------------------------------------------ */
orc_net *orc_parse_cfg(const char *path)
{
    int nsec = 0, i;
    section *secs = read_sections(path, &nsec), *s;
    orc_net *net;
    int h, w, c, inputs, batch;
    size_t ws = 0;
    if (!secs) return NULL;
    if (strcmp(secs->type, "[net]") && strcmp(secs->type, "[network]")) {
        fprintf(stderr, "First section must be [net] or [network]\n");
        return NULL;
    }
    net = calloc(1, sizeof *net);
    net->n = nsec - 1;
    net->layers = calloc(net->n > 0 ? net->n : 1, sizeof(orc_layer));
    /* parser.c:504-523 parse_net_options */
    net->batch = opt_int(secs, "batch", 1);
    net->subdivisions = opt_int(secs, "subdivisions", 1);
    { int ts = opt_int(secs, "time_steps", 1); net->batch /= net->subdivisions; net->batch *= ts; }
    net->h = opt_int(secs, "height", 0);
    net->w = opt_int(secs, "width", 0);
    net->c = opt_int(secs, "channels", 0);
    net->inputs = opt_int(secs, "inputs", net->h * net->w * net->c);
    h = net->h; w = net->w; c = net->c; inputs = net->inputs; batch = net->batch;

    for (s = secs->next, i = 0; s; s = s->next, ++i) {
        orc_layer *l = &net->layers[i];
        const char *t = s->type;
        l->batch = batch;
        if (!strcmp(t, "[convolutional]") || !strcmp(t, "[conv]")) {
            /* parser.c:139-170, convolutional_layer.c:182-235 */
            int pad = opt_int(s, "pad", 0), padding = opt_int(s, "padding", 0);
            l->kind = ORC_CONV;
            l->n = opt_int(s, "filters", 1);
            l->size = opt_int(s, "size", 1);
            l->stride = opt_int(s, "stride", 1);
            if (pad) padding = l->size / 2;
            l->pad = padding;
            l->activation = activation_from_name(opt(s, "activation") ? opt(s, "activation") : "logistic");
            l->batch_normalize = opt_int(s, "batch_normalize", 0);
            l->flipped = opt_int(s, "flipped", 0);
            if (opt_int(s, "binary", 0)) { fprintf(stderr, "oracle: binary=1 convolutions are out of scope (the reference's CPU path corrupts their weights)\n"); return NULL; }
            l->xnor = opt_int(s, "xnor", 0);
            if (!(h && w && c)) { fprintf(stderr, "Layer before convolutional layer must output image.\n"); return NULL; }
            l->h = h; l->w = w; l->c = c;
            l->out_h = (h + 2 * l->pad - l->size) / l->stride + 1;
            l->out_w = (w + 2 * l->pad - l->size) / l->stride + 1;
            l->out_c = l->n;
            l->outputs = l->out_h * l->out_w * l->out_c;
            l->inputs = h * w * c;
            l->weights = calloc((size_t)c * l->n * l->size * l->size, sizeof(float));
            l->biases = calloc(l->n, sizeof(float));
            if (l->batch_normalize) {
                int k;
                l->scales = calloc(l->n, sizeof(float));
                for (k = 0; k < l->n; ++k) l->scales[k] = 1;
                l->rolling_mean = calloc(l->n, sizeof(float));
                l->rolling_variance = calloc(l->n, sizeof(float));
            }
            { size_t need = (size_t)l->out_h * l->out_w * l->size * l->size * c; if (need > ws) ws = need; }
        } else if (!strcmp(t, "[maxpool]") || !strcmp(t, "[max]")) {
            /* parser.c:359-374, maxpool_layer.c:21-52 */
            l->kind = ORC_MAXPOOL;
            l->stride = opt_int(s, "stride", 1);
            l->size = opt_int(s, "size", l->stride);
            l->pad = opt_int(s, "padding", (l->size - 1) / 2);
            l->h = h; l->w = w; l->c = c;
            l->out_w = (w + 2 * l->pad) / l->stride;
            l->out_h = (h + 2 * l->pad) / l->stride;
            l->out_c = c;
            l->outputs = l->out_h * l->out_w * l->out_c;
            l->inputs = h * w * c;
        } else if (!strcmp(t, "[route]")) {
            /* parser.c:450-489, route_layer.c:6-37 */
            const char *ls = opt(s, "layers");
            int n = 1, k;
            const char *p;
            if (!ls) { fprintf(stderr, "Route Layer must specify input layers\n"); return NULL; }
            for (p = ls; *p; ++p) if (*p == ',') ++n;
            l->kind = ORC_ROUTE;
            l->n = n;
            l->input_layers = calloc(n, sizeof(int));
            l->input_sizes = calloc(n, sizeof(int));
            p = ls;
            for (k = 0; k < n; ++k) {
                int idx = atoi(p);
                const char *comma = strchr(p, ',');
                p = comma ? comma + 1 : p;
                if (idx < 0) idx = i + idx;
                l->input_layers[k] = idx;
                l->input_sizes[k] = net->layers[idx].outputs;
                l->outputs += l->input_sizes[k];
            }
            l->inputs = l->outputs;
            {
                orc_layer *first = &net->layers[l->input_layers[0]];
                l->out_w = first->out_w; l->out_h = first->out_h; l->out_c = first->out_c;
                for (k = 1; k < n; ++k) {
                    orc_layer *nx = &net->layers[l->input_layers[k]];
                    if (nx->out_w == first->out_w && nx->out_h == first->out_h) l->out_c += nx->out_c;
                    else l->out_h = l->out_w = l->out_c = 0;
                }
            }
        } else if (!strcmp(t, "[reorg]")) {
            /* parser.c:343-357, reorg_layer.c:7-43 */
            l->kind = ORC_REORG;
            l->stride = opt_int(s, "stride", 1);
            l->reverse = opt_int(s, "reverse", 0);
            l->h = h; l->w = w; l->c = c;
            if (l->reverse) { l->out_w = w * l->stride; l->out_h = h * l->stride; l->out_c = c / (l->stride * l->stride); }
            else { l->out_w = w / l->stride; l->out_h = h / l->stride; l->out_c = c * (l->stride * l->stride); }
            l->outputs = l->out_h * l->out_w * l->out_c;
            l->inputs = h * w * c;
        } else if (!strcmp(t, "[region]")) {
            /* parser.c:236-285, region_layer.c:14-51 */
            const char *a, *tf, *mf;
            int k;
            l->kind = ORC_REGION;
            l->coords = opt_int(s, "coords", 4);
            l->classes = opt_int(s, "classes", 20);
            l->n = opt_int(s, "num", 1);
            l->softmax = opt_int(s, "softmax", 0);
            l->classfix = opt_int(s, "classfix", 0);
            l->h = h; l->w = w;
            l->outputs = h * w * l->n * (l->classes + l->coords + 1);
            l->inputs = l->outputs;
            if (l->outputs != inputs) { fprintf(stderr, "oracle: region outputs %d != inputs %d\n", l->outputs, inputs); return NULL; }
            l->biases = calloc(l->n * 2, sizeof(float));
            for (k = 0; k < l->n * 2; ++k) l->biases[k] = .5f;
            tf = opt(s, "tree"); if (tf) l->tree = orc_read_tree(tf);
            mf = opt(s, "map"); if (mf) l->map = read_int_lines(mf, &l->map_n);
            a = opt(s, "anchors");
            if (a) {
                int n = 1; const char *p;
                for (p = a; *p; ++p) if (*p == ',') ++n;
                p = a;
                for (k = 0; k < n; ++k) {
                    const char *comma;
                    l->biases[k] = (float)atof(p);
                    comma = strchr(p, ',');
                    p = comma ? comma + 1 : p;
                }
            }
        } else if (!strcmp(t, "[avgpool]") || !strcmp(t, "[avg]")) {
            /* parser.c:376-388, avgpool_layer.c:5-31 */
            l->kind = ORC_AVGPOOL;
            l->h = h; l->w = w; l->c = c;
            l->out_w = 1; l->out_h = 1; l->out_c = c;
            l->outputs = c;
            l->inputs = h * w * c;
        } else if (!strcmp(t, "[softmax]") || !strcmp(t, "[soft]")) {
            /* parser.c:226-234, softmax_layer.c:10-33 */
            const char *tf;
            l->kind = ORC_SOFTMAX;
            l->groups = opt_int(s, "groups", 1);
            l->temperature = opt_float(s, "temperature", 1);
            l->inputs = inputs; l->outputs = inputs;
            tf = opt(s, "tree"); if (tf) l->tree = orc_read_tree(tf);
        } else if (!strcmp(t, "[cost]")) {
            /* parser.c:309-317, cost_layer.c:32-55: no-op at inference (cost_layer.c:75) */
            l->kind = ORC_COST;
            l->inputs = inputs; l->outputs = inputs;
        } else if (!strcmp(t, "[connected]") || !strcmp(t, "[conn]")) {
            /* parser.c:214-224, connected_layer.c:13-89 */
            int k;
            l->kind = ORC_CONNECTED;
            l->outputs = opt_int(s, "output", 1);
            l->activation = activation_from_name(opt(s, "activation") ? opt(s, "activation") : "logistic");
            l->batch_normalize = opt_int(s, "batch_normalize", 0);
            l->inputs = inputs;
            l->h = 1; l->w = 1; l->c = inputs;
            l->out_h = 1; l->out_w = 1; l->out_c = l->outputs;
            l->n = l->outputs;
            l->weights = calloc((size_t)l->outputs * inputs, sizeof(float));
            l->biases = calloc(l->outputs, sizeof(float));
            if (l->batch_normalize) {
                l->scales = calloc(l->outputs, sizeof(float));
                for (k = 0; k < l->outputs; ++k) l->scales[k] = 1;
                l->rolling_mean = calloc(l->outputs, sizeof(float));
                l->rolling_variance = calloc(l->outputs, sizeof(float));
            }
        } else if (!strcmp(t, "[dropout]")) {
            /* parser.c:389-397,658-661: at inference the layer's output IS the previous layer's output */
            l->kind = ORC_DROPOUT;
            l->inputs = l->outputs = inputs;
            l->h = l->out_h = h; l->w = l->out_w = w; l->c = l->out_c = c;
        } else if (!strcmp(t, "[detection]")) {
            /* parser.c:285-307, detection_layer.c:14-46 */
            l->kind = ORC_DETECTION;
            l->coords = opt_int(s, "coords", 1);
            l->classes = opt_int(s, "classes", 1);
            l->n = opt_int(s, "num", 1);
            l->side = opt_int(s, "side", 7);
            l->softmax = opt_int(s, "softmax", 0);
            l->sqrt_ = opt_int(s, "sqrt", 0);
            l->inputs = l->outputs = inputs;
            l->w = l->h = l->side;
            if (l->side * l->side * ((1 + l->coords) * l->n + l->classes) != inputs) { fprintf(stderr, "oracle: detection layer size mismatch\n"); return NULL; }
        } else if (!strcmp(t, "[shortcut]")) {
            /* parser.c:415-430, shortcut_layer.c:7-36: l.w/h/c = the `from` layer's output shape, out = this input's */
            const char *fs = opt(s, "from");
            int idx = fs ? atoi(fs) : 0;
            if (!fs) { fprintf(stderr, "oracle: shortcut needs from=\n"); return NULL; }
            if (idx < 0) idx = i + idx;
            if (idx < 0 || idx >= i) { fprintf(stderr, "oracle: shortcut from=%d out of range\n", idx); return NULL; }
            l->kind = ORC_SHORTCUT;
            l->n = idx;                                     /* the reference's l.index */
            l->w = net->layers[idx].out_w; l->h = net->layers[idx].out_h; l->c = net->layers[idx].out_c;
            l->out_w = w; l->out_h = h; l->out_c = c;
            l->outputs = w * h * c;
            l->inputs = l->outputs;
            l->activation = activation_from_name(opt(s, "activation") ? opt(s, "activation") : "linear");
        } else if (!strcmp(t, "[crop]")) {
            /* parser.c:319-341, crop_layer.c:16-46: the inference forward is a centred window, optionally 2x-1 */
            l->kind = ORC_CROP;
            l->out_h = opt_int(s, "crop_height", 1);
            l->out_w = opt_int(s, "crop_width", 1);
            l->noadjust = opt_int(s, "noadjust", 0);
            if (!(h && w && c)) { fprintf(stderr, "Layer before crop layer must output image.\n"); return NULL; }
            l->h = h; l->w = w; l->c = c; l->out_c = c;
            l->inputs = h * w * c;
            l->outputs = l->out_h * l->out_w * l->out_c;
        } else if (!strcmp(t, "[local]")) {
            /* parser.c:118-137, local_layer.c:10-93: out = (in - (pad ? 1 : size)) / stride + 1; one filter bank per location */
            l->kind = ORC_LOCAL;
            l->n = opt_int(s, "filters", 1);
            l->size = opt_int(s, "size", 1);
            l->stride = opt_int(s, "stride", 1);
            l->pad = opt_int(s, "pad", 0);
            l->activation = activation_from_name(opt(s, "activation") ? opt(s, "activation") : "logistic");
            if (!(h && w && c)) { fprintf(stderr, "Layer before local layer must output image.\n"); return NULL; }
            l->h = h; l->w = w; l->c = c;
            l->out_h = (h - (l->pad ? 1 : l->size)) / l->stride + 1;
            l->out_w = (w - (l->pad ? 1 : l->size)) / l->stride + 1;
            l->out_c = l->n;
            l->outputs = l->out_h * l->out_w * l->out_c;
            l->inputs = h * w * c;
            l->weights = calloc((size_t)c * l->n * l->size * l->size * l->out_h * l->out_w, sizeof(float));
            l->biases = calloc(l->outputs, sizeof(float));
            { size_t need = (size_t)l->out_h * l->out_w * l->size * l->size * c; if (need > ws) ws = need; }
            {   /* im2col_cpu derives its own column count (im2col.c:20-21); the layer is only meaningful when both agree */
                int hc = (h + 2 * l->pad - l->size) / l->stride + 1, wc = (w + 2 * l->pad - l->size) / l->stride + 1;
                if (hc != l->out_h || wc != l->out_w) { fprintf(stderr, "oracle: [local] size/pad combination with mismatched im2col grid\n"); return NULL; }
            }
        } else if (!strcmp(t, "[batchnorm]")) {
            /* parser.c:409-413, batchnorm_layer.c:5-57 */
            int k;
            l->kind = ORC_BATCHNORM;
            l->h = l->out_h = h; l->w = l->out_w = w; l->c = l->out_c = c;
            l->inputs = l->outputs = h * w * c;
            l->n = c;
            l->scales = calloc(c, sizeof(float));
            for (k = 0; k < c; ++k) l->scales[k] = 1;
            l->rolling_mean = calloc(c, sizeof(float));
            l->rolling_variance = calloc(c, sizeof(float));
        } else {
            fprintf(stderr, "oracle: layer type %s is outside the hot path\n", t);
            return NULL;
        }
        l->dontload = opt_int(s, "dontload", 0);
        l->dontloadscales = opt_int(s, "dontloadscales", 0);
        if (l->kind == ORC_DROPOUT && i > 0) l->output = net->layers[i - 1].output;      /* parser.c:660 */
        else l->output = calloc((size_t)l->outputs * batch > 0 ? (size_t)l->outputs * batch : 1, sizeof(float));
        h = l->out_h; w = l->out_w; c = l->out_c; inputs = l->outputs;
    }
    free_sections(secs);
    net->workspace_floats = ws;
    net->workspace = calloc(ws ? ws : 1, sizeof(float));
    return net;
}

void orc_free_net(orc_net *net)
{
    int i;
    if (!net) return;
    for (i = 0; i < net->n; ++i) {
        orc_layer *l = &net->layers[i];
        free(l->weights); free(l->biases); free(l->scales); free(l->rolling_mean); free(l->rolling_variance);
        if (l->kind != ORC_DROPOUT) free(l->output);      /* a dropout layer's output is the previous layer's */
        free(l->input_layers); free(l->input_sizes); free(l->map);
        if (l->tree) { free(l->tree->parent); free(l->tree->group); free(l->tree->leaf); free(l->tree->group_size); free(l->tree->group_offset); free(l->tree); }
    }
    free(net->layers);
    free(net->workspace);
    free(net);
}

/* ------------------------------------------------------------------ */
/* load_weights_upto (parser.c:1009-1082), load_convolutional_weights  */
/* (parser.c:963-1006), transpose_matrix (parser.c:884)                */
/* ------------------------------------------------------------------ */
int orc_load_weights(orc_net *net, const char *path)
{
    FILE *fp = fopen(path, "rb");
    int32_t major, minor, revision;
    int i;
    if (!fp) { fprintf(stderr, "oracle: cannot open weights %s\n", path); return -1; }
    if (fread(&major, 4, 1, fp) != 1 || fread(&minor, 4, 1, fp) != 1 || fread(&revision, 4, 1, fp) != 1) { fclose(fp); return -1; }
    if (major * 10 + minor >= 2) { uint64_t s = 0; if (fread(&s, 8, 1, fp) != 1) { fclose(fp); return -1; } net->seen = s; }
    else { int32_t s = 0; if (fread(&s, 4, 1, fp) != 1) { fclose(fp); return -1; } net->seen = (uint64_t)(int64_t)s; }
    for (i = 0; i < net->n; ++i) {
        orc_layer *l = &net->layers[i];
        size_t num, got = 0;
        if (l->dontload) continue;
        if (l->kind == ORC_CONNECTED) {          /* parser.c:897-913 (the transpose flag of :1035 applies to v>1000 files only) */
            got += fread(l->biases, sizeof(float), l->outputs, fp);
            got += fread(l->weights, sizeof(float), (size_t)l->outputs * l->inputs, fp);
            if (l->batch_normalize && !l->dontloadscales) {
                got += fread(l->scales, sizeof(float), l->outputs, fp);
                got += fread(l->rolling_mean, sizeof(float), l->outputs, fp);
                got += fread(l->rolling_variance, sizeof(float), l->outputs, fp);
            }
            continue;
        }
        if (l->kind == ORC_BATCHNORM) {          /* parser.c:921-931 */
            got += fread(l->scales, sizeof(float), l->c, fp);
            got += fread(l->rolling_mean, sizeof(float), l->c, fp);
            got += fread(l->rolling_variance, sizeof(float), l->c, fp);
            continue;
        }
        if (l->kind == ORC_LOCAL) {              /* parser.c:1068-1078 */
            got += fread(l->biases, sizeof(float), l->outputs, fp);
            got += fread(l->weights, sizeof(float), (size_t)l->size * l->size * l->c * l->n * l->out_h * l->out_w, fp);
            continue;
        }
        if (l->kind != ORC_CONV) continue;
        num = (size_t)l->n * l->c * l->size * l->size;
        got += fread(l->biases, sizeof(float), l->n, fp);
        if (l->batch_normalize && !l->dontloadscales) {
            got += fread(l->scales, sizeof(float), l->n, fp);
            got += fread(l->rolling_mean, sizeof(float), l->n, fp);
            got += fread(l->rolling_variance, sizeof(float), l->n, fp);
        }
        got += fread(l->weights, sizeof(float), num, fp);
        (void)got;   /* the reference ignores short reads too */
        if (l->flipped) {
            int rows = l->c * l->size * l->size, cols = l->n, x, y;
            float *tr = calloc((size_t)rows * cols, sizeof(float));
            for (x = 0; x < rows; ++x) for (y = 0; y < cols; ++y) tr[(size_t)y * rows + x] = l->weights[(size_t)x * cols + y];
            memcpy(l->weights, tr, (size_t)rows * cols * sizeof(float));
            free(tr);
        }
    }
    fclose(fp);
    return 0;
}

/* ------------------------------------------------------------------ */
/* primitives                                                          */
/* ------------------------------------------------------------------ */

/* im2col.c:3-37: column matrix [c*k*k][oh*ow], zero outside the image */
static void im2col(const float *im, int channels, int height, int width, int ksize, int stride, int pad, float *col)
{
    int hc = (height + 2 * pad - ksize) / stride + 1;
    int wc = (width + 2 * pad - ksize) / stride + 1;
    int rows = channels * ksize * ksize, r;
#pragma omp parallel for
    for (r = 0; r < rows; ++r) {
        int kw = r % ksize, kh = (r / ksize) % ksize, ch = r / ksize / ksize, y, x;
        for (y = 0; y < hc; ++y) {
            int iy = kh + y * stride - pad;
            float *dst = col + ((size_t)r * hc + y) * wc;
            for (x = 0; x < wc; ++x) {
                int ix = kw + x * stride - pad;
                dst[x] = (iy < 0 || ix < 0 || iy >= height || ix >= width) ? 0.f : im[ix + width * (iy + height * ch)];
            }
        }
    }
}

/* gemm.c:74-88 gemm_nn + gemm.c:141-167 gemm_cpu (TA=TB=0, ALPHA=1, BETA=1):
 * every C[i][j] is a k-ascending chain  c = c + (a*b)  with the product and
 * the sum rounded separately; rows are split over OpenMP threads exactly as
 * gemm.c:156 does, which cannot change any value. */
static void gemm_rows(int M, int N, int K, const float *A, int lda, const float *B, int ldb, float *C, int ldc)
{
    int i;
#pragma omp parallel for
    for (i = 0; i < M; ++i) {
        int k, j;
        float *c = C + (size_t)i * ldc;
        for (k = 0; k < K; ++k) {
            float a = 1.f * A[(size_t)i * lda + k];
            const float *b = B + (size_t)k * ldb;
            for (j = 0; j < N; ++j) c[j] += a * b[j];
        }
    }
}

static float act(float x, int a)   /* activations.h:30-47, activations.c:62-94 */
{
    switch (a) {
    case ACT_LINEAR: return x;
    case ACT_LOGISTIC: return 1. / (1. + exp(-x));
    case ACT_RELU: return x * (x > 0);
    case ACT_LEAKY: return (x > 0) ? x : .1 * x;
    case ACT_RAMP: return x * (x > 0) + .1 * x;
    case ACT_TANH: return (exp(2 * x) - 1) / (exp(2 * x) + 1);
    case ACT_ELU: return (x >= 0) * x + (x < 0) * (exp(x) - 1);
    case ACT_HARDTAN: return x < -1 ? -1 : (x > 1 ? 1 : x);
    case ACT_RELIE: return (x > 0) ? x : .01 * x;
    case ACT_PLSE: if (x < -4) return .01 * (x + 4); if (x > 4) return .01 * (x - 4) + 1; return .125 * x + .5;
    case ACT_LOGGY: return 2. / (1. + exp(-x)) - 1;
    case ACT_STAIR: { int n = floor(x); if (n % 2 == 0) return floor(x / 2.); return (x - n) + floor(x / 2.); }
    case ACT_LHTAN: if (x < 0) return .001 * x; if (x > 1) return .001 * (x - 1) + 1; return x;
    }
    return 0;
}

static float logistic(float x) { return 1. / (1. + exp(-x)); }   /* activations.h:35 */

/* convolutional_layer.c:435-474 forward_convolutional_layer, with the
 * inference branch of batchnorm_layer.c:141-144 (normalize_cpu blas.c:115-126,
 * scale_bias convolutional_layer.c:413-423), add_bias :401-411 and
 * activate_array activations.c:95-101 -- each a separate, separately rounded
 * pass over the output. */
static void forward_conv(orc_net *net, orc_layer *l, const float *input)
{
    int m = l->n, k = l->size * l->size * l->c, n = l->out_h * l->out_w, b, f;
    size_t total = (size_t)l->outputs * l->batch, t;
    const float *wts = l->weights;
    float *bw = NULL, *bin = NULL;
    memset(l->output, 0, total * sizeof(float));
    if (l->xnor) {
        /* convolutional_layer.c:443-447: binarize_weights (:37-50: per filter +-mean|w|, the mean a sequential fp32 sum of
         * double fabs values), binarize_cpu on the input (:52-58: +-1) */
        size_t e, ni = (size_t)l->c * l->h * l->w * l->batch;
        bw = malloc((size_t)m * k * sizeof(float));
        bin = malloc(ni * sizeof(float));
        for (f = 0; f < m; ++f) {
            float mean = 0;
            int i;
            for (i = 0; i < k; ++i) mean += fabs(l->weights[(size_t)f * k + i]);
            mean = mean / k;
            for (i = 0; i < k; ++i) bw[(size_t)f * k + i] = (l->weights[(size_t)f * k + i] > 0) ? mean : -mean;
        }
        for (e = 0; e < ni; ++e) bin[e] = (input[e] > 0) ? 1 : -1;
        wts = bw; input = bin;
    }
    for (b = 0; b < l->batch; ++b) {
        im2col(input + (size_t)b * l->c * l->h * l->w, l->c, l->h, l->w, l->size, l->stride, l->pad, net->workspace);
        gemm_rows(m, n, k, wts, k, net->workspace, n, l->output + (size_t)b * n * m, n);
    }
    free(bw); free(bin);
    if (l->batch_normalize) {
#pragma omp parallel for collapse(2)
        for (b = 0; b < l->batch; ++b) for (f = 0; f < m; ++f) {
            float *x = l->output + ((size_t)b * m + f) * n;
            int i;
            for (i = 0; i < n; ++i) x[i] = (x[i] - l->rolling_mean[f]) / (sqrt(l->rolling_variance[f]) + .000001f);
        }
#pragma omp parallel for collapse(2)
        for (b = 0; b < l->batch; ++b) for (f = 0; f < m; ++f) {
            float *x = l->output + ((size_t)b * m + f) * n;
            int i;
            for (i = 0; i < n; ++i) x[i] *= l->scales[f];
        }
    }
#pragma omp parallel for collapse(2)
    for (b = 0; b < l->batch; ++b) for (f = 0; f < m; ++f) {
        float *x = l->output + ((size_t)b * m + f) * n;
        int i;
        for (i = 0; i < n; ++i) x[i] += l->biases[f];
    }
#pragma omp parallel for
    for (t = 0; t < total; ++t) l->output[t] = act(l->output[t], l->activation);
}

/* maxpool_layer.c:79-114: window origin -pad + o*stride, taps outside the
 * image read as -FLT_MAX, strict '>' so the first maximum wins */
void orc_maxpool(const float *in, int batch, int h, int w, int c, int size, int stride, int pad, float *out)
{
    int oh = (h + 2 * pad) / stride, ow = (w + 2 * pad) / stride, bk;
#pragma omp parallel for
    for (bk = 0; bk < batch * c; ++bk) {
        int i, j, n, m;
        for (i = 0; i < oh; ++i) for (j = 0; j < ow; ++j) {
            float mx = -FLT_MAX;
            for (n = 0; n < size; ++n) for (m = 0; m < size; ++m) {
                int ch = -pad + i * stride + n, cw = -pad + j * stride + m;
                int valid = (ch >= 0 && ch < h && cw >= 0 && cw < w);
                float v = valid ? in[cw + w * (ch + h * (size_t)bk)] : -FLT_MAX;
                mx = (v > mx) ? v : mx;
            }
            out[j + ow * (i + oh * (size_t)bk)] = mx;
        }
    }
}

/* blas.c:8-29 reorg_cpu -- index-for-index, including the quirk that the
 * non-reverse [reorg] layer calls it with forward=0 (reorg_layer.c:83) */
void orc_reorg(const float *x, int w, int h, int c, int batch, int stride, int forward, float *out)
{
    int b, i, j, k, out_c = c / (stride * stride);
    for (b = 0; b < batch; ++b) for (k = 0; k < c; ++k) for (j = 0; j < h; ++j) for (i = 0; i < w; ++i) {
        int in_index = i + w * (j + h * (k + c * b));
        int c2 = k % out_c, offset = k / out_c;
        int w2 = i * stride + offset % stride, h2 = j * stride + offset / stride;
        int out_index = w2 + w * stride * (h2 + h * stride * (c2 + out_c * b));
        if (forward) out[out_index] = x[in_index];
        else out[in_index] = x[out_index];
    }
}

/* blas.c:31-46 flatten(forward=1): per batch item [layers][size] -> [size][layers] */
void orc_flatten(float *x, int size, int layers, int batch, int forward)
{
    size_t total = (size_t)size * layers * batch;
    float *swap = calloc(total ? total : 1, sizeof(float));
    int b, c, i;
    for (b = 0; b < batch; ++b) for (c = 0; c < layers; ++c) for (i = 0; i < size; ++i) {
        size_t i1 = (size_t)b * layers * size + (size_t)c * size + i;
        size_t i2 = (size_t)b * layers * size + (size_t)i * layers + c;
        if (forward) swap[i2] = x[i1]; else swap[i1] = x[i2];
    }
    memcpy(x, swap, total * sizeof(float));
    free(swap);
}

/* blas.c:205-221 softmax: max-subtract, exp in double, fp32 running sum in index order */
void orc_softmax(const float *input, int n, float temp, float *output)
{
    int i;
    float sum = 0, largest = -FLT_MAX;
    for (i = 0; i < n; ++i) if (input[i] > largest) largest = input[i];
    for (i = 0; i < n; ++i) {
        float e = exp(input[i] / temp - largest / temp);
        sum += e;
        output[i] = e;
    }
    for (i = 0; i < n; ++i) output[i] /= sum;
}

/* softmax_layer.c:35-47 softmax_tree (batch = 1 form used by region_layer.c:163) */
static void softmax_tree1(float *x, float temp, const orc_tree *t, float *out)
{
    int g, count = 0;
    for (g = 0; g < t->groups; ++g) {
        orc_softmax(x + count, t->group_size[g], temp, out + count);
        count += t->group_size[g];
    }
}

/* region_layer.c:144-177 (CPU build, !train): memcpy, flatten, logistic on
 * objectness, softmax / tree softmax on classes; x,y,w,h stay raw */
static void forward_region(orc_layer *l, const float *input)
{
    int size = l->coords + l->classes + 1, b, i, total = l->h * l->w * l->n;
    memcpy(l->output, input, (size_t)l->outputs * l->batch * sizeof(float));
    orc_flatten(l->output, l->w * l->h, size * l->n, l->batch, 1);
    for (b = 0; b < l->batch; ++b) for (i = 0; i < total; ++i) {
        size_t index = (size_t)size * i + (size_t)b * l->outputs;
        l->output[index + 4] = logistic(l->output[index + 4]);
    }
    if (l->tree) {
#pragma omp parallel for collapse(2)
        for (b = 0; b < l->batch; ++b) for (i = 0; i < total; ++i) {
            size_t index = (size_t)size * i + (size_t)b * l->outputs;
            softmax_tree1(l->output + index + 5, 1, l->tree, l->output + index + 5);
        }
    } else if (l->softmax) {
        for (b = 0; b < l->batch; ++b) for (i = 0; i < total; ++i) {
            size_t index = (size_t)size * i + (size_t)b * l->outputs;
            orc_softmax(l->output + index + 5, l->classes, 1, l->output + index + 5);
        }
    }
}

/* avgpool_layer.c:40-54: sequential fp32 sum over h*w, then one divide */
static void forward_avgpool(orc_layer *l, const float *input)
{
    int b, k, i, hw = l->h * l->w;
    for (b = 0; b < l->batch; ++b) for (k = 0; k < l->c; ++k) {
        int o = k + b * l->c;
        l->output[o] = 0;
        for (i = 0; i < hw; ++i) l->output[o] += input[i + (size_t)hw * (k + (size_t)b * l->c)];
        l->output[o] /= hw;
    }
}

/* softmax_layer.c:49-61 */
static void forward_softmax(orc_layer *l, const float *input)
{
    int inputs = l->inputs / l->groups, batch = l->batch * l->groups, b;
    for (b = 0; b < batch; ++b) {
        if (l->tree) softmax_tree1((float *)input + (size_t)b * inputs, l->temperature, l->tree, l->output + (size_t)b * inputs);
        else orc_softmax(input + (size_t)b * inputs, inputs, l->temperature, l->output + (size_t)b * inputs);
    }
}

/* network.c:145-160 forward_network + :458-473 network_predict + :173-181 get_network_output */
float *orc_predict(orc_net *net, const float *input)
{
    int i, k, b;
    const float *cur = input;
    for (i = 0; i < net->n; ++i) {
        orc_layer *l = &net->layers[i];
        switch (l->kind) {
        case ORC_CONV: forward_conv(net, l, cur); break;
        case ORC_MAXPOOL: orc_maxpool(cur, l->batch, l->h, l->w, l->c, l->size, l->stride, l->pad, l->output); break;
        case ORC_ROUTE: {           /* route_layer.c:73-86 */
            int offset = 0;
            for (k = 0; k < l->n; ++k) {
                const float *src = net->layers[l->input_layers[k]].output;
                int sz = l->input_sizes[k];
                for (b = 0; b < l->batch; ++b)
                    memcpy(l->output + offset + (size_t)b * l->outputs, src + (size_t)b * sz, (size_t)sz * sizeof(float));
                offset += sz;
            }
        } break;
        case ORC_REORG: orc_reorg(cur, l->w, l->h, l->c, l->batch, l->stride, l->reverse ? 1 : 0, l->output); break;
        case ORC_REGION: forward_region(l, cur); break;
        case ORC_AVGPOOL: forward_avgpool(l, cur); break;
        case ORC_SOFTMAX: forward_softmax(l, cur); break;
        case ORC_COST: break;      /* cost_layer.c:75: returns at once without truth */
        case ORC_DROPOUT: break;   /* dropout_layer.c:34: returns at once unless training; output aliases the input */
        case ORC_CONNECTED: {      /* connected_layer.c:141-176: gemm(0,1,...) = gemm_nt (gemm.c:90-106), BN, bias, activation */
            int o, kk, m = l->batch, n = l->outputs, kdim = l->inputs;
            size_t e;
#pragma omp parallel for collapse(2) private(kk)
            for (b = 0; b < m; ++b) for (o = 0; o < n; ++o) {
                float sum = 0;
                const float *ar = cur + (size_t)b * kdim, *br = l->weights + (size_t)o * kdim;
                for (kk = 0; kk < kdim; ++kk) sum += ar[kk] * br[kk];
                l->output[(size_t)b * n + o] = 0 + sum;
            }
            if (l->batch_normalize) {          /* normalize_cpu blas.c:115-126 with spatial 1, scale_bias */
                for (b = 0; b < m; ++b) for (o = 0; o < n; ++o) {
                    size_t idx = (size_t)b * n + o;
                    l->output[idx] = (l->output[idx] - l->rolling_mean[o]) / (sqrt(l->rolling_variance[o]) + .000001f);
                }
                for (b = 0; b < m; ++b) for (o = 0; o < n; ++o) l->output[(size_t)b * n + o] *= l->scales[o];
            }
            for (b = 0; b < m; ++b) for (o = 0; o < n; ++o) l->output[(size_t)b * n + o] += l->biases[o];
            for (e = 0; e < (size_t)m * n; ++e) l->output[e] = act(l->output[e], l->activation);
        } break;
        case ORC_DETECTION: {      /* detection_layer.c:49-66 (inference part): copy, per-cell class softmax */
            int locations = l->side * l->side, loc;
            memcpy(l->output, cur, (size_t)l->outputs * l->batch * sizeof(float));
            if (l->softmax)
                for (b = 0; b < l->batch; ++b) for (loc = 0; loc < locations; ++loc) {
                    float *pc = l->output + (size_t)b * l->inputs + (size_t)loc * l->classes;
                    orc_softmax(pc, l->classes, 1, pc);
                }
        } break;
        case ORC_CROP: {           /* crop_layer.c:69-105 with !state.train: no flip, centred window, x*2-1 unless noadjust */
            int dh = (l->h - l->out_h) / 2, dw = (l->w - l->out_w) / 2, ch, y, x;
            float scale = l->noadjust ? 1 : 2, trans = l->noadjust ? 0 : -1;
            size_t count = 0;
            for (b = 0; b < l->batch; ++b) for (ch = 0; ch < l->c; ++ch) for (y = 0; y < l->out_h; ++y) for (x = 0; x < l->out_w; ++x)
                l->output[count++] = cur[(x + dw) + (size_t)l->w * ((y + dh) + (size_t)l->h * (ch + (size_t)l->c * b))] * scale + trans;
        } break;
        case ORC_LOCAL: {          /* local_layer.c:95-126: bias copy, per image im2col, per location gemm_nn(n x 1 x k), activation */
            int locations = l->out_h * l->out_w, kdim = l->size * l->size * l->c, j;
            size_t e;
            for (b = 0; b < l->batch; ++b) memcpy(l->output + (size_t)b * l->outputs, l->biases, (size_t)l->outputs * sizeof(float));
            for (b = 0; b < l->batch; ++b) {
                float *out = l->output + (size_t)b * l->outputs;
                im2col(cur + (size_t)b * l->w * l->h * l->c, l->c, l->h, l->w, l->size, l->stride, l->pad, net->workspace);
#pragma omp parallel for
                for (j = 0; j < locations; ++j)
                    gemm_rows(l->n, 1, kdim, l->weights + (size_t)j * kdim * l->n, kdim, net->workspace + j, locations, out + j, locations);
            }
            for (e = 0; e < (size_t)l->outputs * l->batch; ++e) l->output[e] = act(l->output[e], l->activation);
        } break;
        case ORC_BATCHNORM: {      /* batchnorm_layer.c:122-146 (!train): copy, normalize_cpu (blas.c:115-126), scale_bias; no bias term */
            int f, sp = l->out_h * l->out_w;
            for (b = 0; b < l->batch; ++b) for (f = 0; f < l->c; ++f) {
                const float *x = cur + ((size_t)b * l->c + f) * sp;
                float *y = l->output + ((size_t)b * l->c + f) * sp;
                int q;
                for (q = 0; q < sp; ++q) { y[q] = (x[q] - l->rolling_mean[f]) / (sqrt(l->rolling_variance[f]) + .000001f); y[q] *= l->scales[f]; }
            }
        } break;
        case ORC_SHORTCUT: {       /* shortcut_layer.c:38-43: copy, shortcut_cpu (blas.c:57-81), activate_array */
            const float *add = net->layers[l->n].output;
            const int w1 = l->w, h1 = l->h, c1 = l->c, w2 = l->out_w, h2 = l->out_h, c2 = l->out_c;
            int stride = w1 / w2, sample = w2 / w1, minw, minh, minc, x, y, kk;
            size_t e, tot = (size_t)l->outputs * l->batch;
            memcpy(l->output, cur, tot * sizeof(float));
            if (stride < 1) stride = 1;
            if (sample < 1) sample = 1;
            minw = w1 < w2 ? w1 : w2; minh = h1 < h2 ? h1 : h2; minc = c1 < c2 ? c1 : c2;
            for (b = 0; b < l->batch; ++b) for (kk = 0; kk < minc; ++kk) for (y = 0; y < minh; ++y) for (x = 0; x < minw; ++x) {
                size_t out_index = x * sample + (size_t)w2 * (y * sample + (size_t)h2 * (kk + (size_t)c2 * b));
                size_t add_index = x * stride + (size_t)w1 * (y * stride + (size_t)h1 * (kk + (size_t)c1 * b));
                l->output[out_index] += add[add_index];
            }
            for (e = 0; e < tot; ++e) l->output[e] = act(l->output[e], l->activation);
        } break;
        }
        cur = l->output;
    }
    for (i = net->n - 1; i > 0; --i) if (net->layers[i].kind != ORC_COST) break;
    return net->layers[i].output;
}

/* ------------------------------------------------------------------ */
/* decode + NMS                                                        */
/* ------------------------------------------------------------------ */
typedef struct { float x, y, w, h; } orc_box;

/* tree.c:37-51 hierarchy_predictions(only_leaves=0) */
static void hierarchy_predictions(float *p, int n, const orc_tree *t)
{
    int j;
    for (j = 0; j < n; ++j) { int parent = t->parent[j]; if (parent >= 0) p[j] *= p[parent]; }
}

/* region_layer.c:73-85 get_region_box (DOABS=1) and :328-379 get_region_boxes,
 * applied to batch item `b` (the reference reads item 0 only, :331; b>0 is
 * the same code with l.output offset by b*l.outputs -- SURVEY 8b).
 * NOTE: like the reference, the tree branch multiplies the class scores in
 * the layer's output buffer in place.
 * probs is a flat [total][classes] array (the reference's float** rows). */
void orc_get_region_boxes(orc_net *net, int layer, int b, int w, int h, float thresh,
                          float *probs, float *boxes_out, int only_objectness, int use_map)
{
    orc_layer *l = &net->layers[layer];
    float *predictions = l->output + (size_t)b * l->outputs;
    orc_box *boxes = (orc_box *)boxes_out;
    int i, j, n;
    for (i = 0; i < l->w * l->h; ++i) {
        int row = i / l->w, col = i % l->w;
        for (n = 0; n < l->n; ++n) {
            int index = i * l->n + n;
            int p_index = index * (l->classes + 5) + 4;
            int box_index = index * (l->classes + 5);
            int class_index = index * (l->classes + 5) + 5;
            float scale = predictions[p_index];
            float *x = predictions;
            orc_box bx;
            float *pr = probs + (size_t)index * l->classes;
            if (l->classfix == -1 && scale < .5) scale = 0;
            bx.x = (col + logistic(x[box_index + 0])) / l->w;
            bx.y = (row + logistic(x[box_index + 1])) / l->h;
            bx.w = exp(x[box_index + 2]) * l->biases[2 * n] / l->w;
            bx.h = exp(x[box_index + 3]) * l->biases[2 * n + 1] / l->h;
            bx.x *= w; bx.y *= h; bx.w *= w; bx.h *= h;
            boxes[index] = bx;
            if (l->tree) {
                int found = 0;
                hierarchy_predictions(predictions + class_index, l->classes, l->tree);
                if (use_map && l->map) {
                    for (j = 0; j < 200; ++j) {
                        float prob = scale * predictions[class_index + l->map[j]];
                        pr[j] = (prob > thresh) ? prob : 0;
                    }
                } else {
                    for (j = l->classes - 1; j >= 0; --j) {
                        float prob;
                        if (!found && predictions[class_index + j] > .5) found = 1;
                        else predictions[class_index + j] = 0;
                        prob = predictions[class_index + j];
                        pr[j] = (scale > thresh) ? prob : 0;
                    }
                }
            } else {
                for (j = 0; j < l->classes; ++j) {
                    float prob = scale * predictions[class_index + j];
                    pr[j] = (prob > thresh) ? prob : 0;
                }
            }
            if (only_objectness) pr[0] = scale;
        }
    }
}

/* box.c:67-97 overlap / box_intersection / box_union / box_iou (no 0/0 guard) */
static float overlap1(float x1, float w1, float x2, float w2)
{
    float l1 = x1 - w1 / 2, l2 = x2 - w2 / 2;
    float left = l1 > l2 ? l1 : l2;
    float r1 = x1 + w1 / 2, r2 = x2 + w2 / 2;
    float right = r1 < r2 ? r1 : r2;
    return right - left;
}
float orc_box_iou(const float *a, const float *b)
{
    float w = overlap1(a[0], a[2], b[0], b[2]);
    float h = overlap1(a[1], a[3], b[1], b[3]);
    float inter = (w < 0 || h < 0) ? 0 : w * h;
    float uni = a[2] * a[3] + b[2] * b[3] - inter;
    return inter / uni;
}

/* stable merge sort of indices by prob descending: the order box.c:264's
 * qsort + nms_comparator (box.c:239-247, returns 0 on ties) produces with a
 * stable libc sort (glibc 2.35 here): ties keep ascending box index */
static void sort_desc_stable(int *idx, int *tmp, int n, const float *probs, int classes, int k)
{
    int width, i;
    for (width = 1; width < n; width *= 2) {
        for (i = 0; i < n; i += 2 * width) {
            int lo = i, mid = i + width < n ? i + width : n, hi = i + 2 * width < n ? i + 2 * width : n;
            int a = lo, b = mid, o = lo;
            while (a < mid && b < hi) {
                float pa = probs[(size_t)idx[a] * classes + k], pb = probs[(size_t)idx[b] * classes + k];
                if (pb > pa) tmp[o++] = idx[b++]; else tmp[o++] = idx[a++];
            }
            while (a < mid) tmp[o++] = idx[a++];
            while (b < hi) tmp[o++] = idx[b++];
        }
        memcpy(idx, tmp, n * sizeof(int));
    }
}

/* box.c:249-277 do_nms_sort.  As in the reference, the index array persists
 * across classes (each class re-sorts the order the previous class left), so
 * with a stable sort even tied scores come out in the reference's order. */
void orc_do_nms_sort(const float *boxes, float *probs, int total, int classes, float thresh)
{
    int *idx = malloc((total ? total : 1) * sizeof(int)), *tmp = malloc((total ? total : 1) * sizeof(int));
    int i, j, k;
    for (i = 0; i < total; ++i) idx[i] = i;
    for (k = 0; k < classes; ++k) {
        sort_desc_stable(idx, tmp, total, probs, classes, k);
        for (i = 0; i < total; ++i) {
            if (probs[(size_t)idx[i] * classes + k] == 0) continue;
            for (j = i + 1; j < total; ++j)
                if (orc_box_iou(boxes + 4 * (size_t)idx[i], boxes + 4 * (size_t)idx[j]) > thresh)
                    probs[(size_t)idx[j] * classes + k] = 0;
        }
    }
    free(idx); free(tmp);
}

/* box.c:279-298 do_nms (class-agnostic pairwise variant used by demo.c) */
void orc_do_nms(const float *boxes, float *probs, int total, int classes, float thresh)
{
    int i, j, k;
    for (i = 0; i < total; ++i) {
        int any = 0;
        for (k = 0; k < classes; ++k) any = any || (probs[(size_t)i * classes + k] > 0);
        if (!any) continue;
        for (j = i + 1; j < total; ++j) {
            if (orc_box_iou(boxes + 4 * (size_t)i, boxes + 4 * (size_t)j) > thresh) {
                for (k = 0; k < classes; ++k) {
                    if (probs[(size_t)i * classes + k] < probs[(size_t)j * classes + k]) probs[(size_t)i * classes + k] = 0;
                    else probs[(size_t)j * classes + k] = 0;
                }
            }
        }
    }
}

/* utils.c:533-545 max_index: first maximum */
static int max_index(const float *a, int n)
{
    int i, mi = 0; float m;
    if (n <= 0) return -1;
    m = a[0];
    for (i = 1; i < n; ++i) if (a[i] > m) { m = a[i]; mi = i; }
    return mi;
}

/* yolo_v2_class.cpp:221-238: boxes+probs -> bbox_t{x,y,w,h unsigned; prob; obj_id; track_id}
 * out: 7 uint32 words per detection (prob stored as float bits). returns count. */
int orc_detector_bboxes(const float *boxes, const float *probs, int total, int classes, float thresh,
                        int im_w, int im_h, uint32_t *out, int max_out)
{
    int i, cnt = 0;
    for (i = 0; i < total; ++i) {
        const float *b = boxes + 4 * (size_t)i;
        int id = max_index(probs + (size_t)i * classes, classes);
        float prob = probs[(size_t)i * classes + id];
        if (prob > thresh) {
            double dx = (b[0] - b[2] / 2.) * im_w, dy = (b[1] - b[3] / 2.) * im_h;
            uint32_t r[7];
            float fw = b[2] * im_w, fh = b[3] * im_h;
            r[0] = (unsigned int)(dx > 0 ? dx : 0);
            r[1] = (unsigned int)(dy > 0 ? dy : 0);
            r[2] = (unsigned int)fw;
            r[3] = (unsigned int)fh;
            memcpy(&r[4], &prob, 4);
            r[5] = (uint32_t)id;
            r[6] = 0;
            if (cnt < max_out) memcpy(out + 7 * (size_t)cnt, r, sizeof r);
            ++cnt;
        }
    }
    return cnt;
}

/* image.c:33-42 get_color */
static const float orc_colors[6][3] = { {1,0,1}, {0,0,1}, {0,1,1}, {0,1,0}, {1,1,0}, {1,0,0} };
static float get_color(int c, int x, int max)
{
    float ratio = ((float)x / max) * 5;
    int i = floor(ratio), j = ceil(ratio);
    float r;
    ratio -= i;
    r = (1 - ratio) * orc_colors[i][c] + ratio * orc_colors[j][c];
    return r;
}

/* image.c:662-738 draw_detections_test as called by detector.c:558-598
 * test_detector_img: records {x,y,w,h,prob,class,r,g,b} (9 floats) per kept box */
int orc_test_detector_objects(const float *boxes, const float *probs, int total, int classes, float thresh,
                              float *out, int max_out)
{
    int i, cnt = 0;
    for (i = 0; i < total; ++i) {
        int cls = max_index(probs + (size_t)i * classes, classes);
        float prob = probs[(size_t)i * classes + cls];
        if (prob > thresh) {
            int offset = cls * 123457 % classes;
            if (cnt < max_out) {
                float *o = out + 9 * (size_t)cnt;
                o[0] = boxes[4 * (size_t)i]; o[1] = boxes[4 * (size_t)i + 1]; o[2] = boxes[4 * (size_t)i + 2]; o[3] = boxes[4 * (size_t)i + 3];
                o[4] = prob; o[5] = (float)cls;
                o[6] = get_color(2, offset, classes); o[7] = get_color(1, offset, classes); o[8] = get_color(0, offset, classes);
            }
            ++cnt;
        }
    }
    return cnt;
}

/* utils.c:179-193 top_k */
void orc_top_k(const float *a, int n, int k, int *index)
{
    int i, j;
    for (j = 0; j < k; ++j) index[j] = -1;
    for (i = 0; i < n; ++i) {
        int curr = i;
        for (j = 0; j < k; ++j) {
            if (index[j] < 0 || a[curr] > a[index[j]]) { int s = curr; curr = index[j]; index[j] = s; }
            if (curr < 0) break;
        }
    }
}

/* utils.c:420-432 mean_arrays (Detector use_mean smoothing) */
void orc_mean_arrays(const float *const *a, int n, int els, float *avg)
{
    int i, j;
    memset(avg, 0, (size_t)els * sizeof(float));
    for (j = 0; j < n; ++j) for (i = 0; i < els; ++i) avg[i] += a[j][i];
    for (i = 0; i < els; ++i) avg[i] /= n;
}

/* image.c:1950-1992 resize_image: separable align-corners bilinear, two fp32 passes */
void orc_resize_image(const float *im, int iw, int ih, int ic, int w, int h, float *resized)
{
    float *part = calloc((size_t)w * ih * ic + 1, sizeof(float));
    float w_scale = (float)(iw - 1) / (w - 1), h_scale = (float)(ih - 1) / (h - 1);
    int r, c, k;
    for (k = 0; k < ic; ++k) for (r = 0; r < ih; ++r) for (c = 0; c < w; ++c) {
        float val = 0;
        if (c == w - 1 || iw == 1) val = im[(size_t)k * ih * iw + (size_t)r * iw + (iw - 1)];
        else {
            float sx = c * w_scale;
            int ix = (int)sx;
            float dx = sx - ix;
            val = (1 - dx) * im[(size_t)k * ih * iw + (size_t)r * iw + ix] + dx * im[(size_t)k * ih * iw + (size_t)r * iw + ix + 1];
        }
        part[(size_t)k * ih * w + (size_t)r * w + c] = val;
    }
    for (k = 0; k < ic; ++k) for (r = 0; r < h; ++r) {
        float sy = r * h_scale;
        int iy = (int)sy;
        float dy = sy - iy;
        for (c = 0; c < w; ++c) resized[(size_t)k * h * w + (size_t)r * w + c] = (1 - dy) * part[(size_t)k * ih * w + (size_t)iy * w + c];
        if (r == h - 1 || ih == 1) continue;
        for (c = 0; c < w; ++c) resized[(size_t)k * h * w + (size_t)r * w + c] += dy * part[(size_t)k * ih * w + (size_t)(iy + 1) * w + c];
    }
    free(part);
}

/* yolo_v2_class.hpp:94-113 ipl_to_image (out[k][i][j] = data[i*step + j*c + k] / 255.) followed by
 * hpp:133-141 rgbgr_image (planes 0 and 2 exchanged); image.c:2045-2067 load_image_stb is the same
 * conversion without the swap.  Only the first `planes` planes are produced. */
void orc_u8_to_planes(const unsigned char *data, int h, int w, int c, int step, int planes, int swap_rb, float *out)
{
    int i, j, k;
    for (k = 0; k < planes; ++k) {
        int sk = k;
        if (swap_rb && c >= 3) sk = (k == 0) ? 2 : (k == 2 ? 0 : k);
        for (i = 0; i < h; ++i) for (j = 0; j < w; ++j)
            out[((size_t)k * h + i) * w + j] = data[(size_t)i * step + (size_t)j * c + sk] / 255.;
    }
}

/* image.c:1607-1645 letterbox_image / letterbox_image_into: fill = 1 starts from a .5 box (image.c:1637),
 * fill = 0 embeds into what `boxed` already holds; embed_image :1087 through set_pixel :2121 */
void orc_letterbox_image(const float *im, int iw, int ih, int ic, int w, int h, int fill, float *boxed)
{
    int new_w = iw, new_h = ih, k, x, y, dx, dy;
    float *resized;
    if (((float)w / iw) < ((float)h / ih)) { new_w = w; new_h = (ih * w) / iw; }
    else { new_h = h; new_w = (iw * h) / ih; }
    resized = calloc((size_t)new_w * new_h * ic, sizeof(float));
    orc_resize_image(im, iw, ih, ic, new_w, new_h, resized);
    if (fill) { size_t i, n = (size_t)w * h * ic; for (i = 0; i < n; ++i) boxed[i] = .5; }
    dx = (w - new_w) / 2; dy = (h - new_h) / 2;
    for (k = 0; k < ic; ++k) for (y = 0; y < new_h; ++y) for (x = 0; x < new_w; ++x) {
        int X = dx + x, Y = dy + y;
        if (X < 0 || Y < 0 || X >= w || Y >= h) continue;
        boxed[((size_t)k * h + Y) * w + X] = resized[((size_t)k * new_h + y) * new_w + x];
    }
    free(resized);
}

/* Evaluation writers, detector.c:175-243.  kind 0 = print_detector_detections (one file per class,
 * `paths` = classes file names, appended), 1 = print_imagenet_detections (paths[0], numeric id),
 * 2 = print_cocos (paths[0]; image id = digits after the last '_' of `id`, detector.c:169; category ids :23).
 * boxes [total][4] centre form in pixels, probs [total][classes]. */
static const int orc_coco_ids[80] = {1,2,3,4,5,6,7,8,9,10,11,13,14,15,16,17,18,19,20,21,22,23,24,25,27,28,31,32,33,34,35,
    36,37,38,39,40,41,42,43,44,46,47,48,49,50,51,52,53,54,55,56,57,58,59,60,61,62,63,64,65,67,70,72,73,74,75,76,77,78,79,
    80,81,82,84,85,86,87,88,89,90};
int orc_write_detections(int kind, const char **paths, const char *id, int numeric_id, const float *boxes,
                         const float *probs, int total, int classes, int w, int h)
{
    int i, j, nf = kind == 0 ? classes : 1;
    FILE **fps = calloc(nf, sizeof(FILE *));
    for (j = 0; j < nf; ++j) { fps[j] = fopen(paths[j], "a"); if (!fps[j]) return -1; }
    if (kind == 2) { const char *p = strrchr(id, '_'); numeric_id = p ? atoi(p + 1) : 0; }
    for (i = 0; i < total; ++i) {
        const float *b = boxes + (size_t)i * 4;
        float xmin = b[0] - b[2] / 2.;
        float xmax = b[0] + b[2] / 2.;
        float ymin = b[1] - b[3] / 2.;
        float ymax = b[1] + b[3] / 2.;
        if (xmin < 0) xmin = 0;
        if (ymin < 0) ymin = 0;
        if (xmax > w) xmax = w;
        if (ymax > h) ymax = h;
        for (j = 0; j < classes; ++j) {
            float p = probs[(size_t)i * classes + j];
            if (!p) continue;
            if (kind == 0) fprintf(fps[j], "%s %f %f %f %f %f\n", id, p, xmin, ymin, xmax, ymax);
            else if (kind == 1) fprintf(fps[0], "%d %d %f %f %f %f %f\n", numeric_id, j + 1, p, xmin, ymin, xmax, ymax);
            else {
                float bw = xmax - xmin, bh = ymax - ymin;
                fprintf(fps[0], "{\"image_id\":%d, \"category_id\":%d, \"bbox\":[%f, %f, %f, %f], \"score\":%f},\n",
                        numeric_id, orc_coco_ids[j], xmin, ymin, bw, bh, p);
            }
        }
    }
    for (j = 0; j < nf; ++j) fclose(fps[j]);
    free(fps);
    return 0;
}

int orc_last_layer(const orc_net *net);
/* detection_layer.c:222-251 get_detection_boxes on batch item b of the network's last layer */
int orc_get_detection_boxes(orc_net *net, int b, int w, int h, float thresh, int only_objectness, float *boxes, float *probs)
{
    orc_layer *l = &net->layers[orc_last_layer(net)];
    const float *predictions;
    int i, j, n;
    if (l->kind != ORC_DETECTION) return -1;
    predictions = l->output + (size_t)b * l->outputs;
    for (i = 0; i < l->side * l->side; ++i) {
        int row = i / l->side, col = i % l->side;
        for (n = 0; n < l->n; ++n) {
            int index = i * l->n + n;
            int p_index = l->side * l->side * l->classes + i * l->n + n;
            float scale = predictions[p_index];
            int box_index = l->side * l->side * (l->classes + l->n) + (i * l->n + n) * 4;
            boxes[index * 4 + 0] = (predictions[box_index + 0] + col) / l->side * w;
            boxes[index * 4 + 1] = (predictions[box_index + 1] + row) / l->side * h;
            boxes[index * 4 + 2] = pow(predictions[box_index + 2], (l->sqrt_ ? 2 : 1)) * w;
            boxes[index * 4 + 3] = pow(predictions[box_index + 3], (l->sqrt_ ? 2 : 1)) * h;
            for (j = 0; j < l->classes; ++j) {
                int class_index = i * l->classes;
                float prob = scale * predictions[class_index + j];
                probs[(size_t)index * l->classes + j] = (prob > thresh) ? prob : 0;
            }
            if (only_objectness) probs[(size_t)index * l->classes] = scale;
        }
    }
    return 0;
}

/* ------------------------------------------------------------------ */
/* accessors for ctypes                                                */
/* ------------------------------------------------------------------ */
int orc_num_layers(const orc_net *net) { return net->n; }
int orc_batch(const orc_net *net) { return net->batch; }
int orc_inputs(const orc_net *net) { return net->inputs; }
void orc_net_dims(const orc_net *net, int *out4) { out4[0] = net->w; out4[1] = net->h; out4[2] = net->c; out4[3] = net->batch; }
float *orc_layer_output(orc_net *net, int i) { return net->layers[i].output; }
/* info: kind w h c out_w out_h out_c outputs n size stride pad batch_normalize activation classes coords */
void orc_layer_info(const orc_net *net, int i, int *info)
{
    const orc_layer *l = &net->layers[i];
    info[0] = l->kind; info[1] = l->w; info[2] = l->h; info[3] = l->c; info[4] = l->out_w; info[5] = l->out_h; info[6] = l->out_c;
    info[7] = l->outputs; info[8] = l->n; info[9] = l->size; info[10] = l->stride; info[11] = l->pad;
    info[12] = l->batch_normalize; info[13] = l->activation; info[14] = l->classes; info[15] = l->coords;
}
int orc_last_layer(const orc_net *net)
{
    int i;
    for (i = net->n - 1; i > 0; --i) if (net->layers[i].kind != ORC_COST) break;
    return i;
}
float *orc_layer_param(orc_net *net, int i, int which)
{
    orc_layer *l = &net->layers[i];
    switch (which) { case 0: return l->weights; case 1: return l->biases; case 2: return l->scales; case 3: return l->rolling_mean; case 4: return l->rolling_variance; }
    return NULL;
}

/* wall-clock seconds of `iters` forwards (after one warm-up): bench.py cpu_baseline "port" */
double orc_time_predict(orc_net *net, const float *input, int iters)
{
    struct timespec a, b;
    int i;
    orc_predict(net, input);
    clock_gettime(CLOCK_MONOTONIC, &a);
    for (i = 0; i < iters; ++i) orc_predict(net, input);
    clock_gettime(CLOCK_MONOTONIC, &b);
    return (b.tv_sec - a.tv_sec) + 1e-9 * (b.tv_nsec - a.tv_nsec);
}
