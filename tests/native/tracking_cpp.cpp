// Drives Detector::tracking (yolo_v2_class.hpp:77) with box sequences read from a text file and prints the ids it
// hands out, for tests/test_tracking.py to compare with the oracle restatement of yolo_v2_class.cpp:251-303.
//   tracking_cpp <cfg> <sequence.txt> <frames_story>
// sequence: "F n" then n lines "x y w h prob obj_id" per frame.  No weights, no forward: tracking is host logic.
#include <cstdio>
#include <cstdlib>
#include <vector>

#include "yolo_v2_class.hpp"

int main(int argc, char **argv)
{
    if (argc < 4) return 2;
    Detector det(argv[1], "", 0);
    const int story = std::atoi(argv[3]);
    FILE *f = std::fopen(argv[2], "r");
    if (!f) return 2;
    int n;
    char tag;
    while (std::fscanf(f, " %c %d", &tag, &n) == 2 && tag == 'F') {
        std::vector<bbox_t> cur(n);
        for (int i = 0; i < n; ++i) {
            bbox_t &b = cur[i];
            if (std::fscanf(f, "%u %u %u %u %f %u", &b.x, &b.y, &b.w, &b.h, &b.prob, &b.obj_id) != 6) return 3;
            b.track_id = 0;
        }
        std::vector<bbox_t> out = story > 0 ? det.tracking(cur, story) : det.tracking(cur);
        std::printf("F %zu\n", out.size());
        for (const bbox_t &b : out) std::printf("%u %u %u %u %u %u\n", b.x, b.y, b.w, b.h, b.obj_id, b.track_id);
    }
    std::fclose(f);
    return 0;
}
