// lasagna_amd — C++ twin of the reference's batch caller `lasagna align graph.gfa reads.{fa,fq}`
// (/root/reference/src/bin/lasagna.rs:184-276): loads a GFA into a POA graph, aligns every read against the
// fixed graph with Global gap-affine costs (defaults mismatch 4, open 6, extend 2: lasagna.rs:63-106) and prints
// one GAF record per read (src/io/gaf.rs:152-304) with the score in AS:i (lasagna.rs:127-137).
// Where the reference runs `-j` CPU aligner threads, this driver hands the whole batch to the gfx950 library.
//
//   lasagna_amd align [-n MISMATCH] [-g OPEN] [-e EXTEND] [-o OUT] [--mode dense|exact|hybrid] [--device N]
//                     [--dump-graph] GRAPH.gfa READS.fa
#include <cstdio>
#include <cstring>
#include <fstream>
#include <iostream>

#include "../../include/poasta_amd.hpp"

using namespace poasta;

struct SequenceRecord { std::string name, seq; };

static std::vector<SequenceRecord> read_sequences(const std::string& path) {
    std::ifstream f(path);
    if (!f) throw PoastaError("could not open " + path);
    std::vector<SequenceRecord> out;
    std::string line;
    const bool fastq = path.size() > 3 && (path.rfind(".fq") == path.size() - 3 || path.rfind(".fastq") == path.size() - 6);
    if (fastq) {
        while (std::getline(f, line)) {
            if (line.empty() || line[0] != '@') continue;
            SequenceRecord r;
            r.name = line.substr(1, line.find_first_of(" \t") == std::string::npos ? std::string::npos : line.find_first_of(" \t") - 1);
            std::getline(f, r.seq);
            std::getline(f, line);  // +
            std::getline(f, line);  // qualities
            out.push_back(r);
        }
    } else {
        SequenceRecord cur; bool have = false;
        while (std::getline(f, line)) {
            if (!line.empty() && line.back() == '\r') line.pop_back();
            if (line.empty()) continue;
            if (line[0] == '>') {
                if (have) out.push_back(cur);
                cur = SequenceRecord{};
                const size_t sp = line.find_first_of(" \t");
                cur.name = line.substr(1, sp == std::string::npos ? std::string::npos : sp - 1);
                have = true;
            } else if (have) cur.seq += line;
        }
        if (have) out.push_back(cur);
    }
    return out;
}

int main(int argc, char** argv) {
    try {
        if (argc < 2 || std::strcmp(argv[1], "align") != 0) {
            std::fprintf(stderr, "usage: lasagna_amd align [-n 4] [-g 6] [-e 2] [-o out.gaf] [--mode dense|exact|hybrid] [--device 0] [--dump-graph] graph.gfa reads.fa\n");
            return 2;
        }
        int mismatch = 4, open = 6, extend = 2, device = 0;
        bool dump_graph = false;
        std::string out_path, mode = "dense";
        std::vector<std::string> pos;
        for (int i = 2; i < argc; ++i) {
            std::string a = argv[i];
            auto next = [&]() -> std::string { if (i + 1 >= argc) throw PoastaError("missing value for " + a); return argv[++i]; };
            if (a == "-n") mismatch = std::stoi(next());
            else if (a == "-g") open = std::stoi(next());
            else if (a == "-e") extend = std::stoi(next());
            else if (a == "-o") out_path = next();
            else if (a == "--mode") mode = next();
            else if (a == "--device") device = std::stoi(next());
            else if (a == "--dump-graph") dump_graph = true;
            else if (a == "-j") (void)next();  // accepted for CLI compatibility; the GPU batch replaces the thread pool
            else pos.push_back(a);
        }
        if (pos.empty()) throw PoastaError("need graph.gfa");
        io::POAGraphFromGFA gg = io::load_graph_from_gfa(pos[0]);
        io::NodeSegmentResolver resolver(gg.graph, gg.graph_segments);
        if (dump_graph) {
            // machine-readable dump for tests: node count, per-segment (name, start, end, length), adjacency
            std::printf("nodes\t%zu\n", gg.graph.node_count_with_start_and_end());
            for (size_t s = 0; s < gg.graph_segments.names.size(); ++s)
                std::printf("segment\t%s\t%u\t%u\t%zu\n", gg.graph_segments.names[s].c_str(), gg.graph_segments.start_nodes[s],
                            gg.graph_segments.end_nodes[s], gg.graph_segments.segment_lengths[s]);
            for (uint32_t v = 0; v < gg.graph.node_count_with_start_and_end(); ++v) {
                std::printf("node\t%u\t%c\tsucc", v, (char)gg.graph.get_symbol(v));
                for (auto t : gg.graph.successors(v)) std::printf("\t%u", t);
                std::printf("\tpred");
                for (auto t : gg.graph.predecessors(v)) std::printf("\t%u", t);
                auto r = resolver.resolve(v);
                if (r) std::printf("\tseg\t%zu\t%zu", r->first, r->second);
                std::printf("\n");
            }
            if (pos.size() < 2) return 0;
        }
        if (pos.size() < 2) throw PoastaError("need reads.fa");
        auto reads = read_sequences(pos[1]);
        aligner::Mode m = mode == "exact" ? aligner::Mode::Exact : (mode == "hybrid" ? aligner::Mode::Hybrid : aligner::Mode::Dense);
        // GapAffine::new(mismatch, extend, open) — lasagna.rs:193-197; Global is hard-coded there (:256)
        aligner::GapAffine scoring((uint8_t)mismatch, (uint8_t)extend, (uint8_t)open);
        aligner::PoastaAligner<aligner::AffineMinGapCost> al(aligner::AffineMinGapCost(scoring), aligner::AlignmentType::Global, device, m);
        std::vector<std::string> seqs;
        for (auto& r : reads) seqs.push_back(r.seq);
        poa_stats_t st;
        auto results = al.align_batch(gg.graph, seqs, true, &st);
        std::ofstream fout;
        if (!out_path.empty()) { fout.open(out_path); if (!fout) throw PoastaError("could not open " + out_path); }
        std::ostream& os = out_path.empty() ? std::cout : fout;
        for (size_t i = 0; i < reads.size(); ++i) {
            auto rec = io::alignment_to_gaf(gg.graph, gg.graph_segments, reads[i].name, reads[i].seq, results[i].alignment, resolver);
            if (!rec) continue;
            rec->additional_fields.push_back("AS:i:" + std::to_string(results[i].score));
            os << rec->to_string() << "\n";
        }
        std::fprintf(stderr, "lasagna_amd: %zu reads, %llu cells, forward %.3f ms, traceback %.3f ms, exact %.3f ms, flagged %u\n", reads.size(),
                     (unsigned long long)st.cells, st.ms_forward, st.ms_traceback, st.ms_exact, st.n_flagged);
        return 0;
    } catch (const std::exception& ex) {
        std::fprintf(stderr, "lasagna_amd: error: %s\n", ex.what());
        return 1;
    }
}
