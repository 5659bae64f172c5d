// bwa_shim.cpp -- an executable named `bwa` that accepts exactly the command lines the reference emits
// and forwards them to libparasuite_hip.so, so that the UNMODIFIED parasuite.jar drives the GPU path after
// `parasuite setup --parasuite <dir of this binary>` (Main.java:649-654; PARAsuiteMapping.java:35-41).
//
//   bwa index <ref>                                              PARAsuiteMapping.java:48-53
//   bwa parasuite -t T -X mm -p EP -g IP <ref> <fq> -f P.sai     PARAsuiteMapping.java:63-77
//   bwa aln -t T -n mm <ref> <fq> -f P.sai                       BWAMapping.java:51-61
//   bwa samse <ref> P.sai <fq> -f P.sam                          PARAsuiteMapping.java:85-92
//
// Seeding and SAM generation are fused in the library, so `parasuite`/`aln` only record their arguments in
// the .sai file (a text stub only this shim reads; the Java deletes it afterwards, PARAsuiteMapping.java:135-138)
// and `samse` does the work.  Exit status 0/1 and messages on stderr are what Mapping.executeCommand expects
// (Mapping.java:167-172).
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <string>
#include <vector>
#include "../../include/parasuite_hip.h"

static int usage()
{
    std::fprintf(stderr, "Program: bwa (parasuite-hip shim, %s)\nUsage: bwa index|aln|parasuite|samse ...\n", ps_version());
    return 1;
}

int main(int argc, char **argv)
{
    if (argc < 2) return usage();
    const std::string cmd = argv[1];
    if (cmd == "index") {
        if (argc < 3) return usage();
        return ps_index(argv[argc - 1]) ? 1 : 0;
    }
    if (cmd == "aln" || cmd == "parasuite") {
        std::string threads = "1", mm = cmd == "aln" ? "0.04" : "-1", ep, ip, out;
        std::vector<std::string> pos;
        for (int i = 2; i < argc; ++i) {
            const std::string a = argv[i];
            auto val = [&](std::string &dst) { if (i + 1 < argc) dst = argv[++i]; };
            if (a == "-t") val(threads);
            else if (a == "-n" || a == "-X") val(mm);
            else if (a == "-p") val(ep);
            else if (a == "-g") val(ip);
            else if (a == "-f") val(out);
            else if (a.size() > 1 && a[0] == '-' && !(a[1] >= '0' && a[1] <= '9')) { std::string skip; val(skip); }   // other bwa options: accepted, ignored
            else pos.push_back(a);
        }
        if (pos.size() != 2 || out.empty()) { std::fprintf(stderr, "[bwa %s] need <ref> <reads> -f <out.sai>\n", cmd.c_str()); return 1; }
        if (cmd == "parasuite" && ep.empty()) { std::fprintf(stderr, "[bwa parasuite] -p <error profile> is required\n"); return 1; }
        std::ofstream f(out);
        f << "PSSAI1\n" << cmd << '\n' << threads << '\n' << mm << '\n' << ep << '\n' << ip << '\n' << pos[0] << '\n' << pos[1] << '\n';
        return f.good() ? 0 : 1;
    }
    if (cmd == "samse") {
        std::string out; std::vector<std::string> pos;
        for (int i = 2; i < argc; ++i) {
            const std::string a = argv[i];
            if (a == "-f" && i + 1 < argc) out = argv[++i];
            else if (a == "-n" && i + 1 < argc) ++i;          // max alternative hits: the library uses the default 3
            else pos.push_back(a);
        }
        if (pos.size() != 3 || out.empty()) { std::fprintf(stderr, "[bwa samse] need <ref> <in.sai> <reads> -f <out.sam>\n"); return 1; }
        std::ifstream f(pos[1]);
        std::string magic, sub, threads, mm, ep, ip, ref, fq;
        std::getline(f, magic); std::getline(f, sub); std::getline(f, threads); std::getline(f, mm);
        std::getline(f, ep); std::getline(f, ip); std::getline(f, ref); std::getline(f, fq);
        if (magic != "PSSAI1") { std::fprintf(stderr, "[bwa samse] %s was not written by this bwa\n", pos[1].c_str()); return 1; }
        if (ref != pos[0] || fq != pos[2]) { std::fprintf(stderr, "[bwa samse] reference/reads differ from the ones given to `bwa %s`\n", sub.c_str()); return 1; }
        const bool profile = sub == "parasuite";
        return ps_map(std::atoi(threads.c_str()), mm.c_str(), profile ? ep.c_str() : nullptr, profile ? ip.c_str() : nullptr,
                      ref.c_str(), fq.c_str(), out.c_str()) ? 1 : 0;
    }
    return usage();
}
