// poasta_align_amd — C++ twin of the reference's sequential POA construction `poasta align [-I graph.msa.fa] reads.fa -O fasta`
// (/root/reference/src/bin/poasta.rs:163-236, :276-496): the first read seeds the graph, every further read is aligned to the
// graph as it stands (Global, gap-affine, defaults -n 4 -g 6 -e 2: poasta.rs:123-140) and added with
// add_alignment_with_weights (src/graphs/poa.rs:171-321); the result is written as a FASTA MSA (src/io/fasta.rs:69-156).
// Every alignment comes from the gfx950 library (one-query batches: the graph changes after every read, so this caller
// does not shard — DESIGN.md §7); `--mode hybrid` (default) returns the reference's own tie-breaks, which the graph
// update consumes verbatim.
//
//   poasta_align_amd align [-n MISMATCH] [-g OPEN[,OPEN2]] [-e EXTEND[,EXTEND2]] [-H mingap|dijkstra] [-m global|semi-global|ends-free] [-I graph.msa.fa] [-o OUT] [--mode dense|exact|hybrid]
//                          [--device N] [--alignments FILE] READS.fa
//   poasta_align_amd replay ALIGNMENTS.txt          (no GPU: graph update + export from recorded alignments)
#include <chrono>
#include <cstdio>
#include <cstring>
#include <fstream>
#include <iostream>

#include "../../include/poasta_amd.hpp"

using namespace poasta;

static std::vector<std::pair<std::string, std::string>> read_fasta(const std::string& path) {
    std::ifstream f(path);
    if (!f) throw PoastaError("could not open " + path);
    std::vector<std::pair<std::string, std::string>> out;
    std::string line;
    while (std::getline(f, line)) {
        if (!line.empty() && line.back() == '\r') line.pop_back();
        if (line.empty()) continue;
        if (line[0] == '>') {
            const size_t sp = line.find_first_of(" \t");
            out.push_back({line.substr(1, sp == std::string::npos ? std::string::npos : sp - 1), ""});
        } else if (!out.empty()) out.back().second += line;
    }
    return out;
}

// ALIGNMENTS.txt: per read "name<TAB>sequence<TAB>n|-" followed by n lines "rpos qpos" (-1 = None); n == "-": no alignment
static int replay(const std::string& path) {
    std::ifstream f(path);
    if (!f) throw PoastaError("could not open " + path);
    graphs::POAGraph graph;
    std::string first;
    std::getline(f, first);
    if (first.rfind("msa", 0) == 0) {   // "msa<TAB>k": k records "name<TAB>row" seed the graph
        const size_t k = std::stoul(first.substr(4));
        std::vector<std::pair<std::string, std::string>> recs;
        for (size_t i = 0; i < k; ++i) { std::string l; std::getline(f, l); const size_t t = l.find('\t'); recs.push_back({l.substr(0, t), l.substr(t + 1)}); }
        graph = io::load_graph_from_fasta_msa(recs);
        first.clear();
    }
    std::string line = first;
    while (!line.empty() || std::getline(f, line)) {
        if (line.empty()) continue;
        const size_t t1 = line.find('\t'), t2 = line.find('\t', t1 + 1);
        const std::string name = line.substr(0, t1), seq = line.substr(t1 + 1, t2 - t1 - 1), cnt = line.substr(t2 + 1);
        const std::vector<size_t> weights(seq.size(), 1);
        if (cnt == "-") graph.add_alignment_with_weights(name, seq, nullptr, weights);
        else {
            aligner::Alignment aln;
            for (size_t i = 0, n = std::stoul(cnt); i < n; ++i) {
                long long r, q;
                f >> r >> q;
                aligner::AlignedPair ap;
                if (r >= 0) ap.rpos = (graphs::NodeIndex)r;
                if (q >= 0) ap.qpos = (size_t)q;
                aln.push_back(ap);
            }
            std::string rest; std::getline(f, rest);
            graph.add_alignment_with_weights(name, seq, &aln, weights);
        }
        line.clear();
    }
    std::cout << io::poa_graph_to_fasta(graph);
    return 0;
}

int main(int argc, char** argv) {
    try {
        if (argc >= 3 && std::strcmp(argv[1], "replay") == 0) return replay(argv[2]);
        if (argc < 3 || std::strcmp(argv[1], "align") != 0) {
            std::fprintf(stderr, "usage: poasta_align_amd align [-n 4] [-g 6 | 6,24] [-e 2 | 2,1] [-H mingap|dijkstra] [-m global|semi-global|ends-free] [-I graph.msa.fa] [-o out.fa] [--mode dense|exact|hybrid] [--queue-entries-per-cell F] [--device 0] [--alignments file] [--timing file.tsv] reads.fa\n"
                                 "       poasta_align_amd replay alignments.txt\n");
            return 2;
        }
        int mismatch = 4, device = 0;
        float qepc = 0.f;   // replay workspace per table cell (0: the engine's default)
        std::string open_s = "6", extend_s = "2", heuristic = "mingap", span = "global";
        std::string out_path, msa_path, aln_path, timing_path, mode = "hybrid";
        std::vector<std::string> pos;
        for (int i = 2; i < argc; ++i) {
            const std::string a = argv[i];
            auto need = [&](const char* what) { if (i + 1 >= argc) throw PoastaError(std::string("missing value for ") + what); return std::string(argv[++i]); };
            if (a == "-n") mismatch = std::stoi(need("-n"));
            else if (a == "-g") open_s = need("-g");            // "6" or, two-piece model, "6,24" (poasta.rs:122-131)
            else if (a == "-e") extend_s = need("-e");
            else if (a == "-H" || a == "--heuristic") heuristic = need("-H");
            else if (a == "-m" || a == "--alignment-span") span = need("-m");
            else if (a == "-o") out_path = need("-o");
            else if (a == "-I" || a == "--graph") msa_path = need("-I");
            else if (a == "--mode") mode = need("--mode");
            else if (a == "--queue-entries-per-cell") qepc = std::stof(need("--queue-entries-per-cell"));
            else if (a == "--device") device = std::stoi(need("--device"));
            else if (a == "--alignments") aln_path = need("--alignments");
            else if (a == "--timing") timing_path = need("--timing");
            else pos.push_back(a);
        }
        if (pos.size() != 1) throw PoastaError("expected one FASTA of reads");
        const aligner::Mode m = mode == "dense" ? aligner::Mode::Dense : (mode == "exact" ? aligner::Mode::Exact : aligner::Mode::Hybrid);
        const bool mode_dense = mode == "dense";
        bool warned_dense = false;
        graphs::POAGraph graph;
        if (!msa_path.empty()) graph = io::load_graph_from_fasta_msa(read_fasta(msa_path));
        // poasta.rs:275-445: heuristic, span and cost model select the aligner
        auto values = [](const std::string& s) {   // parse_gap_penalties: comma-separated u8 values
            std::vector<int> v; size_t at = 0;
            while (at <= s.size()) { const size_t c = s.find(',', at); v.push_back(std::stoi(s.substr(at, c == std::string::npos ? c : c - at))); if (c == std::string::npos) break; at = c + 1; }
            return v;
        };
        const std::vector<int> go = values(open_s), ge = values(extend_s);
        bool two_piece = go.size() == 2 && ge.size() == 2;
        if (!two_piece && (go.size() != 1 || ge.size() != 1))
            throw PoastaError("Standard affine mode requires exactly 1 value for both gap-open and gap-extend (e.g., -g 6 -e 2)");
        if (two_piece && ge[0] <= ge[1]) {   // poasta.rs:339-342
            std::fprintf(stderr, "Warning: gap_extend1 (%d) should be greater than gap_extend2 (%d) for two-piece model\nUsing standard affine gap model instead.\n", ge[0], ge[1]);
            two_piece = false;
        }
        if (heuristic != "mingap" && heuristic != "dijkstra")
            throw PoastaError(heuristic == "path" ? "heuristic 'path' (PathAware) is not part of this engine: use mingap or dijkstra"
                                                  : "Invalid heuristic type. Valid options are: dijkstra, mingap, path");
        aligner::AlignmentType aln_type = aligner::AlignmentType::Global;
        if (span == "semi-global" || span == "ends-free") aln_type = aligner::AlignmentType::EndsFree();   // both all-Unbounded: poasta.rs:287-300
        else if (span != "global") throw PoastaError("alignment span: global, semi-global or ends-free");
        const bool dij = heuristic == "dijkstra";
        // GapAffine::new(mismatch, extend, open) / GapAffine2Piece::new(mismatch, extend1, open1, extend2, open2): the reference's argument orders
        const aligner::GapAffine c1((uint8_t)mismatch, (uint8_t)ge[0], (uint8_t)go[0]);
        const aligner::GapAffine2Piece c2 = two_piece ? aligner::GapAffine2Piece((uint8_t)mismatch, (uint8_t)ge[0], (uint8_t)go[0], (uint8_t)ge[1], (uint8_t)go[1])
                                                      : aligner::GapAffine2Piece((uint8_t)mismatch, (uint8_t)ge[0], (uint8_t)go[0], (uint8_t)ge[0], (uint8_t)go[0]);
        const aligner::PoastaAligner<aligner::AffineMinGapCost> al_m(aligner::AffineMinGapCost(c1), aln_type, device, m, qepc);
        const aligner::PoastaAligner<aligner::AffineDijkstra> al_d(aligner::AffineDijkstra(c1), aln_type, device, m, qepc);
        const aligner::PoastaAligner<aligner::Affine2PieceMinGapCost> al2_m(aligner::Affine2PieceMinGapCost(c2), aln_type, device, m, qepc);
        const aligner::PoastaAligner<aligner::Affine2PieceDijkstra> al2_d(aligner::Affine2PieceDijkstra(c2), aln_type, device, m, qepc);
        auto align_one = [&](const graphs::POAGraph& gr, const std::string& seq, poa_stats_t* st) {
            if (two_piece) return dij ? al2_d.align_batch(gr, {seq}, true, st).at(0) : al2_m.align_batch(gr, {seq}, true, st).at(0);
            return dij ? al_d.align_batch(gr, {seq}, true, st).at(0) : al_m.align_batch(gr, {seq}, true, st).at(0);
        };
        std::ofstream alog, tlog;
        if (!timing_path.empty()) { tlog.open(timing_path); tlog << "read\tlen\tgraph_nodes\tus_graph_refresh\tus_align_call\tus_h2d\tus_dense_kernels\tus_replay\tus_d2h\tus_graph_update\n"; }
        if (!aln_path.empty()) alog.open(aln_path);
        for (const auto& rec : read_fasta(pos[0])) {
            const std::vector<size_t> weights(rec.second.size(), 1);
            using clk = std::chrono::steady_clock;
            auto us = [](clk::time_point a, clk::time_point b) { return std::chrono::duration<double, std::micro>(b - a).count(); };
            if (graph.is_empty()) {
                graph.add_alignment_with_weights(rec.first, rec.second, nullptr, weights);   // poasta.rs:209-211
                if (alog) alog << rec.first << '\t' << rec.second << "\t-\n";
            } else {
                // per-read latency of BASELINE.json configs[0]'s shape (--timing): the handle's refresh after the last graph update,
                // the alignment call (upload, dense pass, replay of the flagged read, download), the graph update
                const auto t0 = clk::now();
                (void)graph.device_graph();                                                   // poa_graph_update: re-flatten in place
                const auto t1 = clk::now();
                poa_stats_t st{};
                const aligner::AstarResult r = align_one(graph, rec.second, &st);   // poasta.rs:214
                const auto t2 = clk::now();
                std::fprintf(stderr, "Aligned '%s' (len=%zu) - Score: %u, Alignment length: %zu, flags 0x%x\n", rec.first.c_str(),
                             rec.second.size(), r.score, r.alignment.size(), r.flags);
                // The graph update consumes the alignment verbatim (poa.rs:171-321): an alignment that is not the reference's would
                // send every later read down another graph.  A replay that ran out of workspace kept the dense tie-breaks
                // (EXACT_OVERFLOW); where the reference itself would have panicked (REF_PANIC, and the truncated walk that goes
                // with it) there is nothing to reproduce — stop, as the reference does.
                if (r.flags & (POA_FLAG_EXACT_OVERFLOW | POA_FLAG_REF_PANIC)) {
                    std::fprintf(stderr, "poasta_align_amd: read '%s': %s — not adding it to the graph\n", rec.first.c_str(),
                                 (r.flags & POA_FLAG_REF_PANIC) ? "the reference would panic on this input (u32 score wrap)"
                                                                : "the replay ran out of workspace (raise --queue-entries-per-cell)");
                    return 2;
                }
                if (mode_dense && r.flags && !warned_dense) {
                    std::fprintf(stderr, "poasta_align_amd: --mode dense: read '%s' has co-optimal alignments (flags 0x%x); the tie-break taken is the"
                                         " reference's rule on the full table, not necessarily the reference's own — use --mode hybrid for its graph\n", rec.first.c_str(), r.flags);
                    warned_dense = true;
                }
                if (alog) {
                    alog << rec.first << '\t' << rec.second << '\t' << r.alignment.size() << '\n';
                    for (const auto& ap : r.alignment)
                        alog << (ap.rpos ? (long long)*ap.rpos : -1LL) << ' ' << (ap.qpos ? (long long)*ap.qpos : -1LL) << '\n';
                }
                graph.add_alignment_with_weights(rec.first, rec.second, &r.alignment, weights);   // poasta.rs:226
                const auto t3 = clk::now();
                if (tlog) tlog << rec.first << '\t' << rec.second.size() << '\t' << graph.node_count_with_start_and_end() << '\t' << us(t0, t1) << '\t' << us(t1, t2) << '\t'
                               << st.ms_h2d * 1e3 << '\t' << (st.ms_forward + st.ms_traceback) * 1e3 << '\t' << st.ms_exact * 1e3 << '\t' << st.ms_d2h * 1e3 << '\t' << us(t2, t3) << '\n';
            }
        }
        const std::string fasta = io::poa_graph_to_fasta(graph);
        if (out_path.empty()) std::cout << fasta;
        else { std::ofstream o(out_path); o << fasta; }
        return 0;
    } catch (const std::exception& ex) {
        std::fprintf(stderr, "poasta_align_amd: %s\n", ex.what());
        return 1;
    }
}
