/*
 * Darknet .cfg / .data reading and network construction (host side).
 *
 * Accepts the reference's cfg grammar verbatim (src_yolo2/parser.c:702-735
 * read_cfg, src_yolo2/utils.c:230 strip, :263 fgetl, src_yolo2/option_list.c):
 * every blank/tab/CR/LF is removed from each line before parsing, '[' opens a
 * section, lines starting with '#', ';' or empty are skipped, the first
 * occurrence of a key wins, and batch is divided by subdivisions
 * (parser.c:504-514).  Layer geometry follows the reference constructors
 * (convolutional_layer.c:75-83,182-235; maxpool_layer.c:21-52; route_layer.c:6-37
 * + parser.c:450-489; reorg_layer.c:7-43; region_layer.c:14-51 + parser.c:236-285;
 * avgpool_layer.c:5-31; softmax_layer.c:10-33; cost_layer.c:32-55).
 *
 * Sections built: the YOLOv2 / Darknet-19 forward path ([convolutional] [maxpool] [route] [reorg] [region] [avgpool]
 * [softmax], [cost] which does nothing at inference) plus the heads SURVEY 8(f)-4 admits ([shortcut], the YOLOv1
 * head [connected] [dropout] [detection]) and [crop] [local] [batchnorm]; any other section ([rnn], [gru], ...)
 * is an error rather than a silent skip.  No device memory is touched here:
 * HBM buffers are planned at the first predict (y2_engine.c).
 */
#include <stdarg.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <math.h>
#include "y2_internal.h"

int gpu_index = 0;

/* ------------------------------------------------------------------ */
/* errors                                                              */
/* ------------------------------------------------------------------ */
static char g_err[1024] = "";
static int g_mode = 0, g_flag = 0;

const char *y2_last_error(void) { return g_err; }
int y2_error_mode(void) { return g_mode; }
int y2_failed(void) { int f = g_flag; g_flag = 0; return f; }
/* exported for language bindings: 1 = never exit(), record the message instead */
void y2_set_error_mode(int mode) { g_mode = mode; }

void y2_fail(const char *fmt, ...)
{
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof g_err, fmt, ap);
    va_end(ap);
    g_flag = 1;
    if (g_mode == 0) {                      /* utils.c:195-200: report and leave the process */
        fprintf(stderr, "%s\n", g_err);
        exit(-1);
    }
}

/* record a message without ever leaving the process (the C face of the C++ Detector turns exceptions into return codes) */
void y2_set_error_(const char *msg)
{
    snprintf(g_err, sizeof g_err, "%s", msg ? msg : "");
    g_flag = 1;
}

/* for bindings: did the last legacy call fail (and reset the flag); struct sizes for layout checks */
int y2_failed_and_clear(void) { return y2_failed(); }
size_t y2_sizeof_layer(void) { return sizeof(layer); }
size_t y2_sizeof_network(void) { return sizeof(network); }
size_t y2_offsetof_layer_dev(void) { return offsetof(layer, dev); }

void error(const char *s) { y2_fail("%s", s); }

void file_error(char *s)                    /* utils.c:208-213: note the reference exits with status 0 */
{
    if (g_mode == 0) {
        fprintf(stderr, "Couldn't open file: %s\n", s);
        exit(0);
    }
    y2_fail("Couldn't open file: %s", s);
}

/* ------------------------------------------------------------------ */
/* lines and lists                                                     */
/* ------------------------------------------------------------------ */
char *y2_fgetl(FILE *fp)
{
    size_t cap = 512, len = 0;
    char *line;
    int ch;
    if (feof(fp)) return NULL;
    line = malloc(cap);
    while ((ch = fgetc(fp)) != EOF) {
        if (ch == '\n') { line[len] = 0; return line; }
        if (len + 2 > cap) { cap *= 2; line = realloc(line, cap); }
        line[len++] = (char)ch;
    }
    if (len == 0) { free(line); return NULL; }
    line[len] = 0;
    return line;
}

void y2_strip(char *s)
{
    char *w = s;
    for (; *s; ++s) {
        char c = *s;
        if (c == ' ' || c == '\t' || c == '\n' || c == '\r') continue;
        *w++ = c;
    }
    *w = 0;
}

static list *new_list(void) { return calloc(1, sizeof(list)); }

static void list_push(list *l, void *val)
{
    node *n = calloc(1, sizeof(node));
    n->val = val;
    n->prev = l->back;
    if (l->back) l->back->next = n; else l->front = n;
    l->back = n;
    ++l->size;
}

void free_list(list *l)
{
    node *n;
    if (!l) return;
    n = l->front;
    while (n) { node *nx = n->next; free(n); n = nx; }
    free(l);
}

static int add_option(char *line, list *options)   /* option_list.c:35-52 */
{
    char *eq = strchr(line, '=');
    size_t len = strlen(line);
    kvp *p;
    if (eq && (size_t)(eq - line) == len - 1) return 0;    /* "key=" */
    p = calloc(1, sizeof(kvp));
    p->key = line;
    if (eq) { *eq = 0; p->val = eq + 1; }
    list_push(options, p);
    return 1;
}

list *read_data_cfg(char *filename)          /* option_list.c:7-33 */
{
    FILE *fp = fopen(filename, "r");
    list *options;
    char *line;
    int nu = 0;
    if (!fp) { file_error(filename); return NULL; }
    options = new_list();
    while ((line = y2_fgetl(fp)) != NULL) {
        ++nu;
        y2_strip(line);
        if (line[0] == 0 || line[0] == '#' || line[0] == ';') { free(line); continue; }
        if (!add_option(line, options)) {
            fprintf(stderr, "Config file error line %d, could parse: %s\n", nu, line);
            free(line);
        }
    }
    fclose(fp);
    return options;
}

char *option_find(list *l, char *key)        /* option_list.c:74-86 */
{
    node *n;
    for (n = l->front; n; n = n->next) {
        kvp *p = n->val;
        if (strcmp(p->key, key) == 0) { p->used = 1; return p->val; }
    }
    return NULL;
}
char *option_find_str(list *l, char *key, char *def)
{
    char *v = option_find(l, key);
    if (v) return v;
    if (def) fprintf(stderr, "%s: Using default '%s'\n", key, def);
    return def;
}
int option_find_int(list *l, char *key, int def)
{
    char *v = option_find(l, key);
    if (v) return atoi(v);
    fprintf(stderr, "%s: Using default '%d'\n", key, def);
    return def;
}
int option_find_int_quiet(list *l, char *key, int def) { char *v = option_find(l, key); return v ? atoi(v) : def; }
float option_find_float_quiet(list *l, char *key, float def) { char *v = option_find(l, key); return v ? (float)atof(v) : def; }
float option_find_float(list *l, char *key, float def)
{
    char *v = option_find(l, key);
    if (v) return (float)atof(v);
    fprintf(stderr, "%s: Using default '%lf'\n", key, def);
    return def;
}

char **get_labels(char *filename)            /* data.c:474-480: one label per line */
{
    FILE *fp = fopen(filename, "r");
    char **names = NULL, *line;
    int n = 0;
    if (!fp) { file_error(filename); return NULL; }
    while ((line = y2_fgetl(fp)) != NULL) {
        names = realloc(names, (n + 2) * sizeof(char *));
        names[n++] = line;
        names[n] = NULL;
    }
    fclose(fp);
    return names;
}

image **load_alphabet(void) { return NULL; }

int *read_map(char *filename)                /* utils.c:17-30 */
{
    FILE *fp = fopen(filename, "r");
    int n = 0, *map = NULL;
    char *line;
    if (!fp) { file_error(filename); return NULL; }
    while ((line = y2_fgetl(fp)) != NULL) {
        map = realloc(map, (n + 1) * sizeof(int));
        map[n++] = atoi(line);
        free(line);
    }
    fclose(fp);
    if (n < 200) {                           /* get_region_boxes reads 200 entries (region_layer.c:351) */
        map = realloc(map, 200 * sizeof(int));
        for (; n < 200; ++n) map[n] = 0;
    }
    return map;
}

tree *read_tree(char *filename)              /* tree.c:53-101 */
{
    FILE *fp = fopen(filename, "r");
    tree *t;
    char *line;
    int last_parent = -1, in_group = 0, groups = 0, n = 0, i;
    if (!fp) { file_error(filename); return NULL; }
    t = calloc(1, sizeof(tree));
    while ((line = y2_fgetl(fp)) != NULL) {
        char *id = calloc(256, 1);
        int parent = -1;
        sscanf(line, "%255s %d", id, &parent);
        free(line);
        t->parent = realloc(t->parent, (n + 1) * sizeof(int));
        t->name = realloc(t->name, (n + 1) * sizeof(char *));
        t->group = realloc(t->group, (n + 1) * sizeof(int));
        t->parent[n] = parent;
        t->name[n] = id;
        if (parent != last_parent) {         /* siblings are contiguous: a new parent closes the running group */
            t->group_offset = realloc(t->group_offset, (groups + 1) * sizeof(int));
            t->group_size = realloc(t->group_size, (groups + 1) * sizeof(int));
            t->group_offset[groups] = n - in_group;
            t->group_size[groups] = in_group;
            ++groups;
            in_group = 0;
            last_parent = parent;
        }
        t->group[n] = groups;
        ++n;
        ++in_group;
    }
    fclose(fp);
    t->group_offset = realloc(t->group_offset, (groups + 1) * sizeof(int));
    t->group_size = realloc(t->group_size, (groups + 1) * sizeof(int));
    t->group_offset[groups] = n - in_group;
    t->group_size[groups] = in_group;
    ++groups;
    t->n = n;
    t->groups = groups;
    t->leaf = calloc(n > 0 ? n : 1, sizeof(int));
    for (i = 0; i < n; ++i) t->leaf[i] = 1;
    for (i = 0; i < n; ++i) if (t->parent[i] >= 0 && t->parent[i] < n) t->leaf[t->parent[i]] = 0;
    return t;
}

/* ------------------------------------------------------------------ */
/* cfg sections                                                        */
/* ------------------------------------------------------------------ */
typedef struct { char *type; list *options; } section;

static list *read_sections(char *filename)
{
    FILE *fp = fopen(filename, "r");
    list *sections;
    section *cur = NULL;
    char *line;
    int nu = 0;
    if (!fp) { file_error(filename); return NULL; }
    sections = new_list();
    while ((line = y2_fgetl(fp)) != NULL) {
        ++nu;
        y2_strip(line);
        switch (line[0]) {
        case '[':
            cur = calloc(1, sizeof(section));
            cur->type = line;
            cur->options = new_list();
            list_push(sections, cur);
            break;
        case 0: case '#': case ';':
            free(line);
            break;
        default:
            if (!cur || !add_option(line, cur->options)) {
                fprintf(stderr, "Config file error line %d, could parse: %s\n", nu, line);
                free(line);
            }
        }
    }
    fclose(fp);
    return sections;
}

static void free_section(section *s)
{
    node *n = s->options->front;
    while (n) { kvp *p = n->val; free(p->key); free(p); n = n->next; }
    free_list(s->options);
    free(s->type);
    free(s);
}

static void report_unused(list *options)     /* option_list.c:62-72 */
{
    node *n;
    for (n = options->front; n; n = n->next) {
        kvp *p = n->val;
        if (!p->used) fprintf(stderr, "Unused field: '%s = %s'\n", p->key, p->val);
    }
}

static ACTIVATION activation_by_name(const char *s)   /* activations.c get_activation */
{
    static const struct { const char *n; ACTIVATION a; } tab[] = {
        {"logistic", LOGISTIC}, {"loggy", LOGGY}, {"relu", RELU}, {"elu", ELU}, {"relie", RELIE}, {"plse", PLSE},
        {"hardtan", HARDTAN}, {"lhtan", LHTAN}, {"linear", LINEAR}, {"ramp", RAMP}, {"leaky", LEAKY},
        {"tanh", TANH}, {"stair", STAIR},
    };
    size_t i;
    for (i = 0; i < sizeof tab / sizeof tab[0]; ++i) if (strcmp(s, tab[i].n) == 0) return tab[i].a;
    fprintf(stderr, "Couldn't find activation function %s, going with ReLU\n", s);
    return RELU;
}

static int count_commas(const char *s) { int n = 1; for (; *s; ++s) if (*s == ',') ++n; return n; }

static const char *next_item(const char *p) { const char *c = strchr(p, ','); return c ? c + 1 : p + strlen(p); }

typedef struct { int batch, inputs, h, w, c, index; } shape;

static int is_type(const char *t, const char *a, const char *b) { return strcmp(t, a) == 0 || (b && strcmp(t, b) == 0); }

/* the training keys of [net] / [region] are consumed quietly so option_unused stays meaningful */
static void touch(list *o, const char *const *keys) { for (; *keys; ++keys) option_find(o, (char *)*keys); }

static layer make_conv(list *o, shape p)
{
    layer l;
    int pad, padding;
    char *act;
    memset(&l, 0, sizeof l);
    l.type = CONVOLUTIONAL;
    l.n = option_find_int(o, "filters", 1);
    l.size = option_find_int(o, "size", 1);
    l.stride = option_find_int(o, "stride", 1);
    pad = option_find_int_quiet(o, "pad", 0);
    padding = option_find_int_quiet(o, "padding", 0);
    if (pad) padding = l.size / 2;                       /* parser.c:146 */
    l.pad = padding;
    act = option_find_str(o, "activation", "logistic");
    l.activation = activation_by_name(act);
    l.batch_normalize = option_find_int_quiet(o, "batch_normalize", 0);
    l.binary = option_find_int_quiet(o, "binary", 0);
    l.xnor = option_find_int_quiet(o, "xnor", 0);
    l.flipped = option_find_int_quiet(o, "flipped", 0);
    option_find(o, "dot");
    /* binary=1: the reference's CPU forward swaps the binarized copy in only at the END of the layer
     * (convolutional_layer.c:473), i.e. it convolves with the real weights once and with zeros afterwards -- nothing to
     * be compatible with.  xnor=1 (binarized weights and input, :443-447) is implemented. */
    if (l.binary) { y2_fail("binary=1 convolutions are not supported (the reference's CPU path corrupts their weights after the first call)"); return l; }
    if (!(p.h && p.w && p.c)) { y2_fail("Layer before convolutional layer must output image."); return l; }
    if (l.n <= 0 || l.size <= 0 || l.stride <= 0) { y2_fail("bad convolutional geometry"); return l; }
    l.batch = p.batch; l.h = p.h; l.w = p.w; l.c = p.c;
    l.out_h = (l.h + 2 * l.pad - l.size) / l.stride + 1;
    l.out_w = (l.w + 2 * l.pad - l.size) / l.stride + 1;
    l.out_c = l.n;
    l.outputs = l.out_h * l.out_w * l.out_c;
    l.inputs = l.w * l.h * l.c;
    l.weights = calloc((size_t)l.c * l.n * l.size * l.size, sizeof(float));
    l.biases = calloc(l.n, sizeof(float));
    if (l.batch_normalize) {
        int i;
        l.scales = calloc(l.n, sizeof(float));
        for (i = 0; i < l.n; ++i) l.scales[i] = 1;
        l.rolling_mean = calloc(l.n, sizeof(float));
        l.rolling_variance = calloc(l.n, sizeof(float));
    }
    l.workspace_size = (size_t)l.out_h * l.out_w * l.size * l.size * l.c * sizeof(float);
    fprintf(stderr, "conv  %5d %2d x%2d /%2d  %4d x%4d x%4d   ->  %4d x%4d x%4d\n", l.n, l.size, l.size, l.stride,
            l.w, l.h, l.c, l.out_w, l.out_h, l.out_c);
    return l;
}

static layer make_maxpool(list *o, shape p)
{
    layer l;
    memset(&l, 0, sizeof l);
    l.type = MAXPOOL;
    l.stride = option_find_int(o, "stride", 1);
    l.size = option_find_int(o, "size", l.stride);
    l.pad = option_find_int_quiet(o, "padding", (l.size - 1) / 2);
    if (!(p.h && p.w && p.c)) { y2_fail("Layer before maxpool layer must output image."); return l; }
    if (l.size <= 0 || l.stride <= 0) { y2_fail("bad maxpool geometry"); return l; }
    l.batch = p.batch; l.h = p.h; l.w = p.w; l.c = p.c;
    l.out_w = (l.w + 2 * l.pad) / l.stride;
    l.out_h = (l.h + 2 * l.pad) / l.stride;
    l.out_c = l.c;
    l.outputs = l.out_h * l.out_w * l.out_c;
    l.inputs = l.h * l.w * l.c;
    fprintf(stderr, "max          %d x %d / %d  %4d x%4d x%4d   ->  %4d x%4d x%4d\n", l.size, l.size, l.stride,
            l.w, l.h, l.c, l.out_w, l.out_h, l.out_c);
    return l;
}

static layer make_route(list *o, shape p, network *net)
{
    layer l;
    char *ls = option_find(o, "layers");
    const char *q;
    int k, n;
    memset(&l, 0, sizeof l);
    l.type = ROUTE;
    if (!ls) { y2_fail("Route Layer must specify input layers"); return l; }
    n = count_commas(ls);
    l.n = n;
    l.batch = p.batch;
    l.input_layers = calloc(n, sizeof(int));
    l.input_sizes = calloc(n, sizeof(int));
    fprintf(stderr, "route ");
    for (q = ls, k = 0; k < n; ++k, q = next_item(q)) {
        int idx = atoi(q);
        if (idx < 0) idx = p.index + idx;
        if (idx < 0 || idx >= p.index) { y2_fail("route layer %d refers to layer %d", p.index, idx); return l; }
        l.input_layers[k] = idx;
        l.input_sizes[k] = net->layers[idx].outputs;
        l.outputs += l.input_sizes[k];
        fprintf(stderr, " %d", idx);
    }
    fprintf(stderr, "\n");
    l.inputs = l.outputs;
    {
        layer *first = &net->layers[l.input_layers[0]];
        l.out_w = first->out_w; l.out_h = first->out_h; l.out_c = first->out_c;
        for (k = 1; k < n; ++k) {
            layer *nx = &net->layers[l.input_layers[k]];
            if (nx->out_w == first->out_w && nx->out_h == first->out_h) l.out_c += nx->out_c;
            else l.out_h = l.out_w = l.out_c = 0;
        }
    }
    l.h = l.out_h; l.w = l.out_w; l.c = l.out_c;
    return l;
}

/* parser.c:214-224 parse_connected + connected_layer.c:13-89: a dense layer on the flattened input; weights
 * [outputs][inputs] (row = one output), optional batch-norm over the outputs */
static layer make_connected(list *o, shape p)
{
    layer l;
    int i;
    memset(&l, 0, sizeof l);
    l.type = CONNECTED;
    l.outputs = option_find_int(o, "output", 1);
    l.activation = activation_by_name(option_find_str(o, "activation", "logistic"));
    l.batch_normalize = option_find_int_quiet(o, "batch_normalize", 0);
    if (p.inputs <= 0 || l.outputs <= 0) { y2_fail("bad connected layer geometry"); return l; }
    l.inputs = p.inputs;
    l.batch = p.batch;
    l.h = 1; l.w = 1; l.c = l.inputs;
    l.out_h = 1; l.out_w = 1; l.out_c = l.outputs;
    l.n = l.outputs; l.size = 1; l.stride = 1; l.pad = 0;        /* = a 1x1 convolution over a 1x1 image */
    l.weights = calloc((size_t)l.outputs * l.inputs, sizeof(float));
    l.biases = calloc(l.outputs, sizeof(float));
    if (l.batch_normalize) {
        l.scales = calloc(l.outputs, sizeof(float));
        for (i = 0; i < l.outputs; ++i) l.scales[i] = 1;
        l.rolling_mean = calloc(l.outputs, sizeof(float));
        l.rolling_variance = calloc(l.outputs, sizeof(float));
    }
    fprintf(stderr, "connected                            %4d  ->  %4d\n", l.inputs, l.outputs);
    return l;
}

/* parser.c:319-341 parse_crop + crop_layer.c:16-46.  At inference the layer is a centred window of the input,
 * mapped to [-1,1] unless noadjust (crop_layer.c:69-105); flip/angle/saturation/exposure only act in training. */
static layer make_crop(list *o, shape p)
{
    layer l;
    memset(&l, 0, sizeof l);
    l.type = CROP;
    l.out_h = option_find_int(o, "crop_height", 1);
    l.out_w = option_find_int(o, "crop_width", 1);
    l.flip = option_find_int(o, "flip", 0);
    l.angle = option_find_float(o, "angle", 0);
    l.saturation = option_find_float(o, "saturation", 1);
    l.exposure = option_find_float(o, "exposure", 1);
    if (!(p.h && p.w && p.c)) { y2_fail("Layer before crop layer must output image."); return l; }
    l.noadjust = option_find_int_quiet(o, "noadjust", 0);
    l.shift = option_find_float(o, "shift", 0);
    l.batch = p.batch;
    l.h = p.h; l.w = p.w; l.c = p.c;
    if (l.out_h <= 0 || l.out_w <= 0 || l.out_h > l.h || l.out_w > l.w) {
        y2_fail("crop layer: a %d x %d window does not fit the %d x %d input", l.out_h, l.out_w, l.h, l.w);
        return l;
    }
    l.scale = (float)l.out_h / l.h;
    l.out_c = p.c;
    l.inputs = l.w * l.h * l.c;
    l.outputs = l.out_w * l.out_h * l.out_c;
    fprintf(stderr, "Crop Layer: %d x %d -> %d x %d x %d image\n", l.h, l.w, l.out_h, l.out_w, l.c);
    return l;
}

/* parser.c:118-137 parse_local + local_layer.c:10-93: a convolution whose filter bank differs per output location;
 * weights [locations][filters][c*size*size], one bias per output value.  pad is a flag: out = (in - (pad ? 1 : size)) /
 * stride + 1, while the taps come from im2col_cpu with `pad` pixels of zero padding (local_layer.c:107-108). */
static layer make_local(list *o, shape p)
{
    layer l;
    int hc, wc;
    memset(&l, 0, sizeof l);
    l.type = LOCAL;
    l.n = option_find_int(o, "filters", 1);
    l.size = option_find_int(o, "size", 1);
    l.stride = option_find_int(o, "stride", 1);
    l.pad = option_find_int(o, "pad", 0);
    l.activation = activation_by_name(option_find_str(o, "activation", "logistic"));
    if (!(p.h && p.w && p.c)) { y2_fail("Layer before local layer must output image."); return l; }
    if (l.n <= 0 || l.size <= 0 || l.stride <= 0 || l.pad < 0) { y2_fail("bad local layer geometry"); return l; }
    l.batch = p.batch;
    l.h = p.h; l.w = p.w; l.c = p.c;
    l.out_h = (l.h - (l.pad ? 1 : l.size)) / l.stride + 1;
    l.out_w = (l.w - (l.pad ? 1 : l.size)) / l.stride + 1;
    /* the reference multiplies an im2col matrix whose grid is (in + 2*pad - size)/stride + 1 as if it had out_h*out_w
     * columns; only configurations where the two agree are meaningful */
    hc = (l.h + 2 * l.pad - l.size) / l.stride + 1; wc = (l.w + 2 * l.pad - l.size) / l.stride + 1;
    if (l.out_h <= 0 || l.out_w <= 0 || hc != l.out_h || wc != l.out_w) {
        y2_fail("local layer: size=%d pad=%d stride=%d gives a %d x %d output but a %d x %d im2col grid", l.size, l.pad, l.stride,
                l.out_h, l.out_w, hc, wc);
        return l;
    }
    l.out_c = l.n;
    l.outputs = l.out_h * l.out_w * l.out_c;
    l.inputs = l.w * l.h * l.c;
    l.weights = calloc((size_t)l.c * l.n * l.size * l.size * l.out_h * l.out_w, sizeof(float));
    l.biases = calloc(l.outputs, sizeof(float));
    if (!l.weights || !l.biases) { y2_fail("out of memory for local layer weights"); return l; }
    fprintf(stderr, "Local Layer: %d x %d x %d image, %d filters -> %d x %d x %d image\n", l.h, l.w, l.c, l.n, l.out_h, l.out_w, l.n);
    return l;
}

/* parser.c:409-413 parse_batchnorm + batchnorm_layer.c:5-57: a standalone normalisation over the input's channels
 * (scales, rolling mean / variance; no bias) */
static layer make_batchnorm(shape p)
{
    layer l;
    int i;
    memset(&l, 0, sizeof l);
    l.type = BATCHNORM;
    if (!(p.h && p.w && p.c)) { y2_fail("Layer before batchnorm layer must output image."); return l; }
    l.batch = p.batch;
    l.h = l.out_h = p.h; l.w = l.out_w = p.w; l.c = l.out_c = p.c;
    l.inputs = l.outputs = p.w * p.h * p.c;
    l.scales = calloc(l.c, sizeof(float));
    for (i = 0; i < l.c; ++i) l.scales[i] = 1;
    l.rolling_mean = calloc(l.c, sizeof(float));
    l.rolling_variance = calloc(l.c, sizeof(float));
    fprintf(stderr, "Batch Normalization Layer: %d x %d x %d image\n", l.w, l.h, l.c);
    return l;
}

/* parser.c:389-397: at inference a no-op whose output IS the previous layer's (parser.c:658-661) */
static layer make_dropout(list *o, shape p)
{
    layer l;
    memset(&l, 0, sizeof l);
    l.type = DROPOUT;
    l.probability = option_find_float(o, "probability", .5f);
    l.inputs = l.outputs = p.inputs;
    l.batch = p.batch;
    l.h = l.out_h = p.h; l.w = l.out_w = p.w; l.c = l.out_c = p.c;
    fprintf(stderr, "dropout       p = %.2f               %4d  ->  %4d\n", l.probability, l.inputs, l.inputs);
    return l;
}

/* parser.c:285-307 parse_detection + detection_layer.c:14-46 (YOLOv1 head) */
static layer make_detection(list *o, shape p)
{
    static const char *const quiet[] = { "coord_scale", "object_scale", "noobject_scale", "class_scale", "jitter", "random",
                                         "reorg", "max", 0 };
    layer l;
    memset(&l, 0, sizeof l);
    l.type = DETECTION;
    l.coords = option_find_int(o, "coords", 1);
    l.classes = option_find_int(o, "classes", 1);
    l.rescore = option_find_int(o, "rescore", 0);
    l.n = option_find_int(o, "num", 1);
    l.side = option_find_int(o, "side", 7);
    l.softmax = option_find_int(o, "softmax", 0);
    l.sqrt = option_find_int(o, "sqrt", 0);
    l.forced = option_find_int(o, "forced", 0);
    touch(o, quiet);
    l.batch = p.batch;
    l.inputs = l.outputs = p.inputs;
    l.w = l.h = l.side;
    if (l.side * l.side * ((1 + l.coords) * l.n + l.classes) != p.inputs) {     /* detection_layer.c:27 asserts this */
        y2_fail("detection layer: side*side*((1+coords)*num+classes) = %d but the input has %d values",
                l.side * l.side * ((1 + l.coords) * l.n + l.classes), p.inputs);
        return l;
    }
    l.truths = l.side * l.side * (1 + l.coords + l.classes);
    fprintf(stderr, "Detection Layer\n");
    return l;
}

/* parser.c:415-430 parse_shortcut + shortcut_layer.c:7-36: l.w/h/c = shape of the `from` layer's output,
 * out = this layer's input shape, l.index = the `from` layer (the reference's field of that name) */
static layer make_shortcut(list *o, shape p, network *net)
{
    layer l, *from;
    char *fs = option_find(o, "from");
    int idx;
    memset(&l, 0, sizeof l);
    l.type = SHORTCUT;
    if (!fs) { y2_fail("Shortcut layer must specify from="); return l; }
    idx = atoi(fs);
    if (idx < 0) idx = p.index + idx;
    if (idx < 0 || idx >= p.index) { y2_fail("shortcut layer %d refers to layer %d", p.index, idx); return l; }
    if (!(p.h && p.w && p.c)) { y2_fail("Layer before shortcut layer must output image."); return l; }
    from = &net->layers[idx];
    if (!(from->out_h && from->out_w && from->out_c)) { y2_fail("shortcut layer %d: layer %d does not output an image", p.index, idx); return l; }
    l.index = idx;
    l.batch = p.batch;
    l.w = from->out_w; l.h = from->out_h; l.c = from->out_c;
    l.out_w = p.w; l.out_h = p.h; l.out_c = p.c;
    l.outputs = p.w * p.h * p.c;
    l.inputs = l.outputs;
    l.activation = activation_by_name(option_find_str(o, "activation", "linear"));
    fprintf(stderr, "Shortcut Layer: %d\n", idx);
    return l;
}

static layer make_reorg(list *o, shape p)
{
    layer l;
    memset(&l, 0, sizeof l);
    l.type = REORG;
    l.stride = option_find_int(o, "stride", 1);
    l.reverse = option_find_int_quiet(o, "reverse", 0);
    if (!(p.h && p.w && p.c)) { y2_fail("Layer before reorg layer must output image."); return l; }
    if (l.stride <= 0) { y2_fail("bad reorg stride"); return l; }
    l.batch = p.batch; l.h = p.h; l.w = p.w; l.c = p.c;
    if (l.reverse) { l.out_w = l.w * l.stride; l.out_h = l.h * l.stride; l.out_c = l.c / (l.stride * l.stride); }
    else { l.out_w = l.w / l.stride; l.out_h = l.h / l.stride; l.out_c = l.c * (l.stride * l.stride); }
    l.outputs = l.out_h * l.out_w * l.out_c;
    l.inputs = l.h * l.w * l.c;
    fprintf(stderr, "reorg              /%2d  %4d x%4d x%4d   ->  %4d x%4d x%4d\n", l.stride, l.w, l.h, l.c,
            l.out_w, l.out_h, l.out_c);
    return l;
}

static layer make_region(list *o, shape p)
{
    static const char *const quiet[] = { "object_scale", "noobject_scale", "class_scale", "coord_scale", 0 };
    layer l;
    char *a, *tf, *mf;
    int i;
    memset(&l, 0, sizeof l);
    l.type = REGION;
    l.coords = option_find_int(o, "coords", 4);
    l.classes = option_find_int(o, "classes", 20);
    l.n = option_find_int(o, "num", 1);
    l.batch = p.batch; l.h = p.h; l.w = p.w;
    l.outputs = l.h * l.w * l.n * (l.classes + l.coords + 1);
    l.inputs = l.outputs;
    l.truths = 30 * 5;
    if (l.coords != 4) { y2_fail("region layer: only coords=4 is supported"); return l; }
    if (l.outputs != p.inputs) { y2_fail("region layer: %d outputs but the previous layer has %d", l.outputs, p.inputs); return l; }
    l.log = option_find_int_quiet(o, "log", 0);
    l.sqrt = option_find_int_quiet(o, "sqrt", 0);
    l.softmax = option_find_int(o, "softmax", 0);
    l.max_boxes = option_find_int_quiet(o, "max", 30);
    l.jitter = option_find_float(o, "jitter", .2f);
    l.rescore = option_find_int_quiet(o, "rescore", 0);
    l.thresh = option_find_float(o, "thresh", .5f);
    l.classfix = option_find_int_quiet(o, "classfix", 0);
    l.absolute = option_find_int_quiet(o, "absolute", 0);
    l.random = option_find_int_quiet(o, "random", 0);
    l.coord_scale = option_find_float_quiet(o, "coord_scale", 1);
    l.object_scale = option_find_float_quiet(o, "object_scale", 1);
    l.noobject_scale = option_find_float_quiet(o, "noobject_scale", 1);
    l.class_scale = option_find_float_quiet(o, "class_scale", 1);
    l.bias_match = option_find_int_quiet(o, "bias_match", 0);
    touch(o, quiet);
    l.biases = calloc(l.n * 2, sizeof(float));
    for (i = 0; i < l.n * 2; ++i) l.biases[i] = .5f;
    l.cost = calloc(1, sizeof(float));
    tf = option_find_str(o, "tree", 0);
    if (tf) l.softmax_tree = read_tree(tf);
    mf = option_find_str(o, "map", 0);
    if (mf) l.map = read_map(mf);
    a = option_find_str(o, "anchors", 0);
    if (a) {
        int n = count_commas(a);
        const char *q = a;
        if (n > 2 * l.n) n = 2 * l.n;                    /* the reference writes past l.biases here */
        for (i = 0; i < n; ++i, q = next_item(q)) l.biases[i] = (float)atof(q);
    }
    if (l.softmax_tree && l.softmax_tree->n != l.classes) {
        y2_fail("region layer: tree has %d nodes but classes=%d (the reference would read out of bounds, tree.c:37)",
                l.softmax_tree->n, l.classes);
        return l;
    }
    fprintf(stderr, "detection\n");
    return l;
}

static layer make_avgpool(shape p)
{
    layer l;
    memset(&l, 0, sizeof l);
    l.type = AVGPOOL;
    if (!(p.h && p.w && p.c)) { y2_fail("Layer before avgpool layer must output image."); return l; }
    l.batch = p.batch; l.h = p.h; l.w = p.w; l.c = p.c;
    l.out_w = 1; l.out_h = 1; l.out_c = l.c;
    l.outputs = l.out_c;
    l.inputs = l.h * l.w * l.c;
    fprintf(stderr, "avg                     %4d x%4d x%4d   ->  %4d\n", l.w, l.h, l.c, l.c);
    return l;
}

static layer make_softmax(list *o, shape p)
{
    layer l;
    char *tf;
    memset(&l, 0, sizeof l);
    l.type = SOFTMAX;
    l.groups = option_find_int_quiet(o, "groups", 1);
    l.temperature = option_find_float_quiet(o, "temperature", 1);
    l.batch = p.batch;
    l.inputs = p.inputs; l.outputs = p.inputs;
    if (l.groups <= 0 || l.inputs % l.groups) { y2_fail("softmax: inputs %d not divisible by groups %d", l.inputs, l.groups); return l; }
    tf = option_find_str(o, "tree", 0);
    if (tf) l.softmax_tree = read_tree(tf);
    fprintf(stderr, "softmax                                        %4d\n", l.inputs);
    return l;
}

static layer make_cost(list *o, shape p)
{
    layer l;
    char *ts;
    memset(&l, 0, sizeof l);
    l.type = COST;
    ts = option_find_str(o, "type", "sse");
    l.cost_type = strcmp(ts, "masked") == 0 ? MASKED : (strcmp(ts, "smooth") == 0 ? SMOOTH : SSE);
    l.scale = option_find_float_quiet(o, "scale", 1);
    l.batch = p.batch;
    l.inputs = p.inputs; l.outputs = p.inputs;
    l.cost = calloc(1, sizeof(float));
    fprintf(stderr, "cost                                           %4d\n", l.inputs);
    return l;
}

network make_network(int n)                  /* network.c:132-143 */
{
    network net;
    memset(&net, 0, sizeof net);
    net.n = n;
    net.layers = calloc(n > 0 ? n : 1, sizeof(layer));
    net.seen = calloc(2, sizeof(int));        /* room for the 64-bit counter of version >= 0.2 files */
    net.gpu_index = gpu_index;
    return net;
}

int y2_out_layer(const network *net)         /* network.c:173-181: last layer that is not [cost] */
{
    int i;
    for (i = net->n - 1; i > 0; --i) if (net->layers[i].type != COST) break;
    return i;
}

network parse_network_cfg(char *filename)    /* parser.c:585-700 */
{
    static const char *const net_quiet[] = {
        "max_crop", "min_crop", "angle", "aspect", "saturation", "exposure", "hue", "policy", "burn_in", "step",
        "scale", "steps", "scales", "gamma", "power", "adam", "B1", "B2", "eps", 0 };
    network net, empty;
    list *sections;
    node *n;
    section *s;
    shape p;
    int count = 0;

    memset(&empty, 0, sizeof empty);
    (void)y2_failed();                        /* a stale failure of an earlier call must not fail this parse */
    sections = read_sections(filename);
    if (!sections) return empty;
    n = sections->front;
    if (!n) { y2_fail("Config file has no sections"); return empty; }
    s = n->val;
    if (!is_type(s->type, "[net]", "[network]")) { y2_fail("First section must be [net] or [network]"); return empty; }

    net = make_network(sections->size - 1);
    /* parser.c:504-523 */
    net.batch = option_find_int(s->options, "batch", 1);
    net.learning_rate = option_find_float(s->options, "learning_rate", .001f);
    net.momentum = option_find_float(s->options, "momentum", .9f);
    net.decay = option_find_float(s->options, "decay", .0001f);
    net.subdivisions = option_find_int(s->options, "subdivisions", 1);
    net.time_steps = option_find_int_quiet(s->options, "time_steps", 1);
    if (net.subdivisions <= 0) net.subdivisions = 1;
    net.batch /= net.subdivisions;
    net.batch *= net.time_steps;
    net.h = option_find_int_quiet(s->options, "height", 0);
    net.w = option_find_int_quiet(s->options, "width", 0);
    net.c = option_find_int_quiet(s->options, "channels", 0);
    net.inputs = option_find_int_quiet(s->options, "inputs", net.h * net.w * net.c);
    net.max_batches = option_find_int_quiet(s->options, "max_batches", 0);
    touch(s->options, net_quiet);
    if (!net.inputs && !(net.h && net.w && net.c)) { y2_fail("No input parameters supplied"); return empty; }
    if (net.batch <= 0) { y2_fail("batch/subdivisions gives a batch of %d", net.batch); return empty; }

    p.h = net.h; p.w = net.w; p.c = net.c; p.inputs = net.inputs; p.batch = net.batch; p.index = 0;
    free_section(s);
    fprintf(stderr, "layer     filters    size              input                output\n");
    for (n = n->next; n; n = n->next, ++count) {
        layer l;
        const char *t;
        s = n->val;
        t = s->type;
        p.index = count;
        fprintf(stderr, "%5d ", count);
        if (is_type(t, "[convolutional]", "[conv]")) l = make_conv(s->options, p);
        else if (is_type(t, "[maxpool]", "[max]")) l = make_maxpool(s->options, p);
        else if (is_type(t, "[route]", NULL)) l = make_route(s->options, p, &net);
        else if (is_type(t, "[reorg]", NULL)) l = make_reorg(s->options, p);
        else if (is_type(t, "[shortcut]", NULL)) l = make_shortcut(s->options, p, &net);
        else if (is_type(t, "[connected]", "[conn]")) l = make_connected(s->options, p);
        else if (is_type(t, "[dropout]", NULL)) l = make_dropout(s->options, p);
        else if (is_type(t, "[detection]", NULL)) l = make_detection(s->options, p);
        else if (is_type(t, "[crop]", NULL)) l = make_crop(s->options, p);
        else if (is_type(t, "[local]", NULL)) l = make_local(s->options, p);
        else if (is_type(t, "[batchnorm]", NULL)) l = make_batchnorm(p);
        else if (is_type(t, "[region]", NULL)) l = make_region(s->options, p);
        else if (is_type(t, "[avgpool]", "[avg]")) l = make_avgpool(p);
        else if (is_type(t, "[softmax]", "[soft]")) { l = make_softmax(s->options, p); net.hierarchy = l.softmax_tree; }
        else if (is_type(t, "[cost]", NULL)) l = make_cost(s->options, p);
        else {
            memset(&l, 0, sizeof l);
            y2_fail("layer type %s is outside the YOLOv2/Darknet-19 forward path this engine implements", t);
        }
        if (y2_failed()) { g_flag = 1; return empty; }
        if (l.type != SHORTCUT) l.index = count;      /* a shortcut keeps the index of its `from` layer there */
        l.dontload = option_find_int_quiet(s->options, "dontload", 0);
        l.dontloadscales = option_find_int_quiet(s->options, "dontloadscales", 0);
        report_unused(s->options);
        net.layers[count] = l;
        free_section(s);
        p.h = l.out_h; p.w = l.out_w; p.c = l.out_c; p.inputs = l.outputs;
    }
    free_list(sections);
    net.outputs = net.layers[y2_out_layer(&net)].outputs;
    if (y2_engine_create(&net) != 0) return empty;
    net.output = NULL;                        /* allocated with the plan; see get_network_output */
    return net;
}
