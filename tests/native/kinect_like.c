/* A caller written the way the reference's Kinect application drives the engine
 * (KinectUtil.cpp:81-92 initialise, :403 per frame; detector.c:558 test_detector_img),
 * compiled against include/ with the reference's own header names and linked to
 * libsr_yolo2.so.  Prints one line per detected object for the test to compare.
 *
 *   kinect_like <cfg> <weights> <frame.bin: c h w int32 header + CHW float32> <thresh> [names]
 */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "network.h"
#include "parser.h"
#include "region_layer.h"
#include "box.h"
#include "cuda.h"
#include "utils.h"
#include "image.h"
#include "option_list.h"
#include "test_detector.h"

int main(int argc, char **argv)
{
    if (argc < 5) { fprintf(stderr, "usage: kinect_like cfg weights frame.bin thresh\n"); return 2; }
    cuda_set_device(0);
    network net = parse_network_cfg(argv[1]);
    load_weights(&net, argv[2]);
    set_batch_network(&net, 1);

    FILE *f = fopen(argv[3], "rb");
    int hdr[3];
    if (!f || fread(hdr, sizeof(int), 3, f) != 3) { fprintf(stderr, "bad frame file\n"); return 2; }
    image im = make_image(hdr[2], hdr[1], hdr[0]);
    if (fread(im.data, sizeof(float), (size_t)im.w * im.h * im.c, f) != (size_t)im.w * im.h * im.c) return 2;
    fclose(f);

    float thresh = (float)atof(argv[4]);
    layer l = net.layers[net.n - 1];
    char **names = calloc(l.classes, sizeof(char *));
    int i;
    for (i = 0; i < l.classes; ++i) { names[i] = malloc(32); snprintf(names[i], 32, "class%d", i); }

    object objs[2048];
    int n = 0, frame;
    for (frame = 0; frame < 2; ++frame) {          /* the application calls this every frame */
        n = 0;
        test_detector_img(names, load_alphabet(), net, im, thresh, objs, &n);
    }
    printf("OBJECTS %d net %dx%d classes %d\n", n, net.w, net.h, l.classes);
    for (i = 0; i < n; ++i)
        printf("OBJ %d %.9g %.9g %.9g %.9g %.9g %s %.9g %.9g %.9g\n", objs[i].objClass, objs[i].x, objs[i].y, objs[i].w,
               objs[i].h, objs[i].prob, objs[i].name, objs[i].boxRGB[0], objs[i].boxRGB[1], objs[i].boxRGB[2]);

    /* the lower-level sequence of detector.c:567-574 through the legacy host-array functions */
    {
        image sized = resize_image(im, net.w, net.h);
        int total = l.w * l.h * l.n, kept = 0, j;
        box *boxes = calloc(total, sizeof(box));
        float **probs = calloc(total, sizeof(float *));
        for (j = 0; j < total; ++j) probs[j] = calloc(l.classes, sizeof(float));
        float *out = network_predict(net, sized.data);
        (void)out;
        get_region_boxes(l, 1, 1, thresh, probs, boxes, 0, 0);
        do_nms_sort(boxes, probs, total, l.classes, 0.1f);
        for (j = 0; j < total; ++j) {
            int c = max_index(probs[j], l.classes);
            if (probs[j][c] > thresh) ++kept;
        }
        printf("LEGACY %d outputs %d\n", kept, get_network_output_size(net));
        free_image(sized);
    }
    free_network(net);
    return 0;
}
