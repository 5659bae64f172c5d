// samtools_shim.cpp -- argv-compatible `samtools` for exactly the command shapes the unmodified PARA-suite jar issues
// after the map step (found through PATH):
//   samtools view -bS -t <ref> <in.sam> -o <out.bam>      PARAsuiteMapping.java:102-111, BWAMapping.java:80-89
//   samtools view -q <mapq> -b <in.bam> -o <out.bam>      PARAsuiteMapping.java:124-133
//   samtools sort [-n] <in.bam> -o <out.bam>              Mapping.java:86-93, 118-126
//   samtools index <in.bam>                               Mapping.java:100-105, 132-137
// Anything else is refused with exit status 1 (this is not samtools).  Work is done by libparasuite_hip.so (host code).
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>
#include "../../include/parasuite_hip.h"

static int usage(const char *why)
{
    std::fprintf(stderr, "[parasuite-hip samtools shim] %s\nsupported: view -bS [-t ref] in.sam -o out.bam | view -q Q -b in.bam -o out.bam | sort [-n] in.bam -o out.bam | index in.bam\n", why);
    return 1;
}

int main(int argc, char **argv)
{
    if (argc < 3) return usage("too few arguments");
    const std::string cmd = argv[1];
    int threads = 8;
    if (const char *e = std::getenv("PARASUITE_THREADS")) threads = std::atoi(e) > 0 ? std::atoi(e) : 8;
    std::vector<std::string> pos; std::string out; int q = 0; bool flag_b = false, flag_S = false, flag_n = false;
    for (int i = 2; i < argc; ++i) {
        const std::string a = argv[i];
        if (a == "-o" && i + 1 < argc) out = argv[++i];
        else if (a == "-q" && i + 1 < argc) q = std::atoi(argv[++i]);
        else if (a == "-t" && i + 1 < argc) ++i;                       // reference list: our SAM carries its @SQ lines
        else if (a == "-@" && i + 1 < argc) threads = std::atoi(argv[++i]);
        else if (a == "-n") flag_n = true;
        else if (a.size() > 1 && a[0] == '-') {
            for (size_t k = 1; k < a.size(); ++k) {
                if (a[k] == 'b') flag_b = true; else if (a[k] == 'S') flag_S = true; else if (a[k] == 'h') {}
                else return usage(("option not supported: " + a).c_str());
            }
        } else pos.push_back(a);
    }
    int rc = 1;
    if (cmd == "view") {
        if (pos.size() != 1 || out.empty() || !flag_b) return usage("view needs -b, one input and -o");
        const bool sam_in = flag_S || (pos[0].size() > 4 && pos[0].compare(pos[0].size() - 4, 4, ".sam") == 0);
        rc = sam_in ? ps_sam_to_bam(pos[0].c_str(), out.c_str(), q, 0, 0, threads, nullptr) : ps_bam_view(pos[0].c_str(), out.c_str(), q, threads, nullptr);
    } else if (cmd == "sort") {
        if (pos.size() != 1 || out.empty()) return usage("sort needs one input and -o");
        rc = ps_bam_sort(pos[0].c_str(), out.c_str(), flag_n ? 1 : 0, threads, nullptr);
    } else if (cmd == "index") {
        if (pos.size() != 1) return usage("index needs one input");
        rc = ps_bam_index(pos[0].c_str(), threads);
    } else return usage(("command not supported: " + cmd).c_str());
    if (rc) std::fprintf(stderr, "[parasuite-hip samtools shim] %s\n", ps_last_error());
    return rc ? 1 : 0;
}
