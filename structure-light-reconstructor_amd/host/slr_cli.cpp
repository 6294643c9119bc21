// slr_cli -- command-line stand-in for MainWindow::startreconstruct (Duke/mainwindow.cpp:562-652): reads a project
// directory in the reference's layout, runs one reconstruction on the GPU and writes reconstruction/<sn>.ply.
//   slr_cli <project> <mode: gray|grayepi|mf> [--sn N] [--scan W H] [--cam W H] [--black T] [--white T] [--color] [--suffix .png|.pgm]
//           [--series K]   (mf only: scans N .. N+K-1, pipelined: PNG decode of the next scan overlaps the GPU work of this one)
//           [--devices 0,1,..]   (with --series: the scans are dealt round-robin to these GPUs, one pipeline per entry)
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <string>

extern "C" int duke_run_project(const char *, int, int, int, int, int, int, int, int, int, const char *, const char *, float *,
                                unsigned char *, unsigned char *, char *, int);
extern "C" int duke_run_series(const char *, int, int, int, int, int, int, int, int, const char *, const char *, float *,
                               unsigned char *, char *, int);
extern "C" int duke_run_series_multi(const char *, int, int, int, int, int, int, int, int, const char *, const char *, const int *, int,
                                     float *, unsigned char *, char *, int);

int main(int argc, char **argv)
{
    if (argc < 3) {
        fprintf(stderr, "usage: %s <project> <gray|grayepi|mf> [--sn N] [--scan W H] [--cam W H] [--black T] [--white T] [--color] [--suffix .png]\n", argv[0]);
        return 2;
    }
    const std::string project = argv[1], m = argv[2];
    const int mode = m == "gray" ? 0 : m == "grayepi" ? 1 : 2;
    // defaults: Duke/Set.ui (scan 1280x1024, camera 1280x1024, blackThreshold 40, whiteThreshold 0)
    int sn = 0, sw = 1280, sh = 1024, cw = 1280, chh = 1024, black = 40, white = 0, color = 0, series = 0;
    std::string suffix = ".png";
    int devices[64], n_devices = 0;
    for (int i = 3; i < argc; i++) {
        if (!strcmp(argv[i], "--sn") && i + 1 < argc) sn = atoi(argv[++i]);
        else if (!strcmp(argv[i], "--scan") && i + 2 < argc) { sw = atoi(argv[++i]); sh = atoi(argv[++i]); }
        else if (!strcmp(argv[i], "--cam") && i + 2 < argc) { cw = atoi(argv[++i]); chh = atoi(argv[++i]); }
        else if (!strcmp(argv[i], "--black") && i + 1 < argc) black = atoi(argv[++i]);
        else if (!strcmp(argv[i], "--white") && i + 1 < argc) white = atoi(argv[++i]);
        else if (!strcmp(argv[i], "--color")) color = 1;
        else if (!strcmp(argv[i], "--series") && i + 1 < argc) series = atoi(argv[++i]);
        else if (!strcmp(argv[i], "--suffix") && i + 1 < argc) suffix = argv[++i];
        else if (!strcmp(argv[i], "--devices") && i + 1 < argc) {
            for (char *tok = strtok(argv[++i], ","); tok && n_devices < 64; tok = strtok(NULL, ",")) devices[n_devices++] = atoi(tok);
        }
    }
    char err[512] = "";
    if (series > 0 && mode == 2) {
        const std::string pre = project + "/reconstruction/";
        const int done = n_devices > 0
            ? duke_run_series_multi(project.c_str(), sn, series, sw, sh, cw, chh, black, white, suffix.c_str(), pre.c_str(), devices, n_devices,
                                    NULL, NULL, err, (int)sizeof err)
            : duke_run_series(project.c_str(), sn, series, sw, sh, cw, chh, black, white, suffix.c_str(), pre.c_str(), NULL, NULL,
                              err, (int)sizeof err);
        printf("wrote %d of %d meshes to %s<sn>.ply\n", done, series, pre.c_str());
        if (done != series) { fprintf(stderr, "series stopped: %s\n", err); return 1; }
        return 0;
    }
    const std::string ply = project + "/reconstruction/" + std::to_string(sn) + ".ply";
    const int ok = duke_run_project(project.c_str(), mode, sn, sw, sh, cw, chh, black, white, color, suffix.c_str(), ply.c_str(),
                                    NULL, NULL, NULL, err, (int)sizeof err);
    if (!ok) { fprintf(stderr, "reconstruction failed: %s\n", err); return 1; }
    printf("wrote %s\n", ply.c_str());
    return 0;
}
