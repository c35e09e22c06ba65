// Driver (ours) around two of the REFERENCE's helper subcommands, compiled from the sources where they lie:
//   ref_tools cov2bed <reference.fa> <context> <bismark.cov> <out.bed>   cov_to_bed_main       (src/app/hifimeth/cov_to_bed.cpp)
//   ref_tools corr <a.cov.bed> <b.cov.bed> ...                           pileup_correlation_main (src/app/hifimeth/pileup_correlation.cpp)
// argv is handed over unchanged: both mains expect argv[1] to be the subcommand name (main.cpp:40-46).
#include <cstdio>
#include <cstring>

int cov_to_bed_main(int argc, char* argv[]);
int pileup_correlation_main(int argc, char* argv[]);

int main(int argc, char* argv[]) {
    if (argc >= 2 && strcmp(argv[1], "cov2bed") == 0) return cov_to_bed_main(argc, argv);
    if (argc >= 2 && strcmp(argv[1], "corr") == 0) return pileup_correlation_main(argc, argv);
    fprintf(stderr, "usage: ref_tools cov2bed|corr ...\n");
    return 2;
}
