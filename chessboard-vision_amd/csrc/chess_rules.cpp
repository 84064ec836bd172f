// Game rules for SURVEY §8 f1 (include/cbv_chess.h): the part of python-chess that game_state.py uses,
// restated from the rules of chess and python-chess's documented behaviour (Board.fen() with
// en_passant="legal", cleaned castling rights, king-move encoding of castling, generation order of
// legal_moves), and GameState.process_occupancy_change (game_state.py:40-195).
// Host C++ only.  The generator is checked against the published perft numbers in tests/test_chess_rules.py.
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "../../include/cbv_chess.h"

namespace {

typedef uint64_t u64;
enum { PAWN = 1, KNIGHT, BISHOP, ROOK, QUEEN, KING };
enum { WK = 1, WQ = 2, BK = 4, BQ = 8 };

inline int file_of(int s) { return s & 7; }
inline int rank_of(int s) { return s >> 3; }
inline u64 bit(int s) { return 1ull << s; }
inline int msb(u64 v) { return 63 - __builtin_clzll(v); }

struct Undo {
    cbv_move m;
    int8_t captured, castling, ep;
    int halfmove;
};

} // namespace

struct cbv_board {
    int8_t sq[64]; // 0 empty, type | 8 for black
    int turn;      // 1 white, 0 black
    int castling, ep, halfmove, fullmove;
    std::vector<Undo> stack;
};

namespace {

inline bool is_white(int p) { return p != 0 && !(p & 8); }
inline bool is_black(int p) { return (p & 8) != 0; }
inline bool own(const cbv_board* b, int p) { return p != 0 && (b->turn ? is_white(p) : is_black(p)); }
inline bool enemy(const cbv_board* b, int p) { return p != 0 && (b->turn ? is_black(p) : is_white(p)); }

const int KN[8][2] = {{1, 2}, {2, 1}, {2, -1}, {1, -2}, {-1, -2}, {-2, -1}, {-2, 1}, {-1, 2}};
const int KG[8][2] = {{1, 0}, {1, 1}, {0, 1}, {-1, 1}, {-1, 0}, {-1, -1}, {0, -1}, {1, -1}};
const int DIAG[4][2] = {{1, 1}, {-1, 1}, {-1, -1}, {1, -1}};
const int ORTH[4][2] = {{1, 0}, {0, 1}, {-1, 0}, {0, -1}};

u64 step_mask(int s, const int (*d)[2], int n)
{
    u64 m = 0;
    for (int i = 0; i < n; i++) {
        int f = file_of(s) + d[i][0], r = rank_of(s) + d[i][1];
        if (f >= 0 && f < 8 && r >= 0 && r < 8) m |= bit(r * 8 + f);
    }
    return m;
}

u64 ray_mask(const cbv_board* b, int s, const int (*d)[2], int n)
{
    u64 m = 0;
    for (int i = 0; i < n; i++) {
        int f = file_of(s) + d[i][0], r = rank_of(s) + d[i][1];
        while (f >= 0 && f < 8 && r >= 0 && r < 8) {
            m |= bit(r * 8 + f);
            if (b->sq[r * 8 + f]) break;
            f += d[i][0];
            r += d[i][1];
        }
    }
    return m;
}

// squares a piece of `type` on `s` attacks (pawns excluded)
u64 attacks_from(const cbv_board* b, int type, int s)
{
    switch (type) {
    case KNIGHT: return step_mask(s, KN, 8);
    case KING: return step_mask(s, KG, 8);
    case BISHOP: return ray_mask(b, s, DIAG, 4);
    case ROOK: return ray_mask(b, s, ORTH, 4);
    case QUEEN: return ray_mask(b, s, DIAG, 4) | ray_mask(b, s, ORTH, 4);
    }
    return 0;
}

// is square s attacked by the side `white` (1) / black (0)?
bool attacked(const cbv_board* b, int s, int white)
{
    const int side = white ? 0 : 8;
    u64 m = step_mask(s, KN, 8);
    while (m) {
        int t = msb(m);
        m &= ~bit(t);
        if (b->sq[t] == (KNIGHT | side)) return true;
    }
    m = step_mask(s, KG, 8);
    while (m) {
        int t = msb(m);
        m &= ~bit(t);
        if (b->sq[t] == (KING | side)) return true;
    }
    // a white pawn on (f +- 1, r - 1) attacks (f, r)
    const int pr = rank_of(s) + (white ? -1 : 1);
    if (pr >= 0 && pr < 8)
        for (int df = -1; df <= 1; df += 2) {
            int f = file_of(s) + df;
            if (f >= 0 && f < 8 && b->sq[pr * 8 + f] == (PAWN | side)) return true;
        }
    m = ray_mask(b, s, DIAG, 4);
    while (m) {
        int t = msb(m);
        m &= ~bit(t);
        if (b->sq[t] == (BISHOP | side) || b->sq[t] == (QUEEN | side)) return true;
    }
    m = ray_mask(b, s, ORTH, 4);
    while (m) {
        int t = msb(m);
        m &= ~bit(t);
        if (b->sq[t] == (ROOK | side) || b->sq[t] == (QUEEN | side)) return true;
    }
    return false;
}

int king_square(const cbv_board* b, int white)
{
    const int k = KING | (white ? 0 : 8);
    for (int s = 63; s >= 0; s--)
        if (b->sq[s] == k) return s;
    return -1;
}

inline cbv_move mk(int from, int to, int promo = 0) { return (cbv_move)(from | (to << 6) | (promo << 12)); }
inline int m_from(cbv_move m) { return m & 63; }
inline int m_to(cbv_move m) { return (m >> 6) & 63; }
inline int m_promo(cbv_move m) { return (m >> 12) & 7; }

// castling rights whose king and rook still stand where they must (python-chess clean_castling_rights)
int clean_castling(const cbv_board* b)
{
    int c = b->castling;
    if (b->sq[4] != KING) c &= ~(WK | WQ);
    if (b->sq[7] != ROOK) c &= ~WK;
    if (b->sq[0] != ROOK) c &= ~WQ;
    if (b->sq[60] != (KING | 8)) c &= ~(BK | BQ);
    if (b->sq[63] != (ROOK | 8)) c &= ~BK;
    if (b->sq[56] != (ROOK | 8)) c &= ~BQ;
    return c;
}

bool is_ep_move(const cbv_board* b, cbv_move m)
{
    const int from = m_from(m), to = m_to(m);
    if (b->ep < 0 || to != b->ep) return false;
    if ((b->sq[from] & 7) != PAWN) return false;
    const int d = to - from;
    if (d != 7 && d != 9 && d != -7 && d != -9) return false;
    return b->sq[to] == 0;
}

bool is_castling_move(const cbv_board* b, cbv_move m)
{
    const int from = m_from(m), to = m_to(m);
    return (b->sq[from] & 7) == KING && abs(file_of(from) - file_of(to)) == 2 && rank_of(from) == rank_of(to);
}

void do_push(cbv_board* b, cbv_move m)
{
    const int from = m_from(m), to = m_to(m), promo = m_promo(m);
    const int piece = b->sq[from], type = piece & 7, side = piece & 8;
    Undo u;
    u.m = m;
    u.castling = (int8_t)b->castling;
    u.ep = (int8_t)b->ep;
    u.halfmove = b->halfmove;
    u.captured = b->sq[to];
    const bool ep_cap = is_ep_move(b, m);
    const bool castle = is_castling_move(b, m);
    const bool zeroing = type == PAWN || b->sq[to] != 0 || ep_cap;
    b->halfmove = zeroing ? 0 : b->halfmove + 1;
    if (!b->turn) b->fullmove++;
    b->ep = -1;
    // castling rights
    if (type == KING) b->castling &= side ? ~(BK | BQ) : ~(WK | WQ);
    if (from == 7 || to == 7) b->castling &= ~WK;
    if (from == 0 || to == 0) b->castling &= ~WQ;
    if (from == 63 || to == 63) b->castling &= ~BK;
    if (from == 56 || to == 56) b->castling &= ~BQ;
    b->sq[from] = 0;
    if (ep_cap) {
        const int victim = to + (side ? 8 : -8);
        u.captured = b->sq[victim];
        b->sq[victim] = 0;
    }
    if (type == PAWN && abs(to - from) == 16) b->ep = (from + to) / 2;
    b->sq[to] = (int8_t)(promo ? (promo | side) : piece);
    if (castle) {
        const int r = rank_of(from) * 8;
        if (file_of(to) == 6) {
            b->sq[r + 5] = b->sq[r + 7];
            b->sq[r + 7] = 0;
        } else {
            b->sq[r + 3] = b->sq[r + 0];
            b->sq[r + 0] = 0;
        }
    }
    b->turn ^= 1;
    b->stack.push_back(u);
}

cbv_move do_pop(cbv_board* b)
{
    if (b->stack.empty()) return CBV_MOVE_NONE;
    const Undo u = b->stack.back();
    b->stack.pop_back();
    b->turn ^= 1;
    const int from = m_from(u.m), to = m_to(u.m), promo = m_promo(u.m);
    int piece = b->sq[to];
    const int side = piece & 8;
    if (promo) piece = PAWN | side;
    b->castling = u.castling;
    b->ep = u.ep;
    b->halfmove = u.halfmove;
    if (!b->turn) b->fullmove--;
    b->sq[from] = (int8_t)piece;
    b->sq[to] = 0;
    // restore captures / en passant / castling rook (judged on the restored position)
    const bool was_ep = (piece & 7) == PAWN && b->ep >= 0 && to == b->ep && file_of(from) != file_of(to);
    if (was_ep) b->sq[to + (side ? 8 : -8)] = u.captured;
    else b->sq[to] = u.captured;
    if ((piece & 7) == KING && abs(file_of(from) - file_of(to)) == 2) {
        const int r = rank_of(from) * 8;
        if (file_of(to) == 6) {
            b->sq[r + 7] = b->sq[r + 5];
            b->sq[r + 5] = 0;
        } else {
            b->sq[r + 0] = b->sq[r + 3];
            b->sq[r + 3] = 0;
        }
    }
    return u.m;
}

// after the side to move played m, is its own king attacked?  (m is pseudo-legal)
bool leaves_king_safe(cbv_board* b, cbv_move m)
{
    const int white = b->turn;
    do_push(b, m);
    const int k = king_square(b, white);
    const bool safe = k < 0 || !attacked(b, k, !white);
    do_pop(b);
    return safe;
}

void add_pawn_move(std::vector<cbv_move>& out, int from, int to)
{
    if (rank_of(to) == 0 || rank_of(to) == 7) {
        out.push_back(mk(from, to, QUEEN));
        out.push_back(mk(from, to, ROOK));
        out.push_back(mk(from, to, BISHOP));
        out.push_back(mk(from, to, KNIGHT));
    } else out.push_back(mk(from, to));
}

void gen_ep(const cbv_board* b, std::vector<cbv_move>& out, u64 from_mask, u64 to_mask)
{
    if (b->ep < 0 || b->sq[b->ep] || !(to_mask & bit(b->ep))) return;
    const int want_rank = b->turn ? 4 : 3;
    const int pr = rank_of(b->ep) + (b->turn ? -1 : 1);
    if (pr != want_rank) return;
    for (int f = file_of(b->ep) + 1; f >= file_of(b->ep) - 1; f -= 2) { // scan_reversed: higher square first
        if (f < 0 || f > 7) continue;
        const int s = pr * 8 + f;
        if ((from_mask & bit(s)) && b->sq[s] == (PAWN | (b->turn ? 0 : 8))) out.push_back(mk(s, b->ep));
    }
}

// python-chess generate_pseudo_legal_moves(from_mask, to_mask), in its order: pieces (high square first,
// targets high first), castling (h side first), pawn captures, single pushes, double pushes, en passant
void gen_pseudo(const cbv_board* b, std::vector<cbv_move>& out, u64 from_mask, u64 to_mask)
{
    u64 ours = 0, theirs = 0;
    for (int s = 0; s < 64; s++) {
        if (own(b, b->sq[s])) ours |= bit(s);
        else if (b->sq[s]) theirs |= bit(s);
    }
    const int side = b->turn ? 0 : 8;
    for (int s = 63; s >= 0; s--) {
        const int p = b->sq[s];
        if (!own(b, p) || (p & 7) == PAWN || !(from_mask & bit(s))) continue;
        u64 t = attacks_from(b, p & 7, s) & ~ours & to_mask;
        while (t) {
            const int to = msb(t);
            t &= ~bit(to);
            out.push_back(mk(s, to));
        }
    }
    // castling
    {
        const int rights = clean_castling(b);
        const int r = b->turn ? 0 : 56, ks = r + 4;
        if ((from_mask & bit(ks)) && b->sq[ks] == (KING | side)) {
            const bool k_right = rights & (b->turn ? WK : BK), q_right = rights & (b->turn ? WQ : BQ);
            if (k_right && (to_mask & bit(r + 6)) && !b->sq[r + 5] && !b->sq[r + 6] && !attacked(b, ks, !b->turn) &&
                !attacked(b, r + 5, !b->turn) && !attacked(b, r + 6, !b->turn))
                out.push_back(mk(ks, r + 6));
            if (q_right && (to_mask & bit(r + 2)) && !b->sq[r + 3] && !b->sq[r + 2] && !b->sq[r + 1] &&
                !attacked(b, ks, !b->turn) && !attacked(b, r + 3, !b->turn) && !attacked(b, r + 2, !b->turn))
                out.push_back(mk(ks, r + 2));
        }
    }
    const int fwd = b->turn ? 8 : -8;
    // pawn captures
    for (int s = 63; s >= 0; s--) {
        if (b->sq[s] != (PAWN | side) || !(from_mask & bit(s))) continue;
        const int tr = rank_of(s) + (b->turn ? 1 : -1);
        if (tr < 0 || tr > 7) continue;
        for (int f = file_of(s) + 1; f >= file_of(s) - 1; f -= 2) { // higher target square first
            if (f < 0 || f > 7) continue;
            const int to = tr * 8 + f;
            if ((theirs & bit(to)) && (to_mask & bit(to))) add_pawn_move(out, s, to);
        }
    }
    // single then double pushes, by target square from high to low
    for (int to = 63; to >= 0; to--) {
        const int from = to - fwd;
        if (from < 0 || from > 63 || b->sq[to] || b->sq[from] != (PAWN | side)) continue;
        if (!(from_mask & bit(from)) || !(to_mask & bit(to))) continue;
        add_pawn_move(out, from, to);
    }
    for (int to = 63; to >= 0; to--) {
        if (rank_of(to) != (b->turn ? 3 : 4)) continue;
        const int mid = to - fwd, from = to - 2 * fwd;
        if (b->sq[to] || b->sq[mid] || b->sq[from] != (PAWN | side)) continue;
        if (!(from_mask & bit(from)) || !(to_mask & bit(to))) continue;
        out.push_back(mk(from, to));
    }
    gen_ep(b, out, from_mask, to_mask);
}

u64 between_mask(int a, int c)
{
    const int df = file_of(c) - file_of(a), dr = rank_of(c) - rank_of(a);
    if (!((df == 0) || (dr == 0) || (abs(df) == abs(dr)))) return 0;
    const int sf = (df > 0) - (df < 0), sr = (dr > 0) - (dr < 0);
    u64 m = 0;
    int f = file_of(a) + sf, r = rank_of(a) + sr;
    while (f != file_of(c) || r != rank_of(c)) {
        m |= bit(r * 8 + f);
        f += sf;
        r += sr;
    }
    return m;
}

u64 line_mask(int a, int c) // the whole line through a and c (python-chess ray), 0 if not aligned
{
    const int df = file_of(c) - file_of(a), dr = rank_of(c) - rank_of(a);
    if (!((df == 0) || (dr == 0) || (abs(df) == abs(dr))) || (df == 0 && dr == 0)) return 0;
    const int sf = (df > 0) - (df < 0), sr = (dr > 0) - (dr < 0);
    u64 m = bit(a);
    for (int dir = -1; dir <= 1; dir += 2) {
        int f = file_of(a) + dir * sf, r = rank_of(a) + dir * sr;
        while (f >= 0 && f < 8 && r >= 0 && r < 8) {
            m |= bit(r * 8 + f);
            f += dir * sf;
            r += dir * sr;
        }
    }
    return m;
}

u64 attackers_of(const cbv_board* b, int s, int white)
{
    u64 m = 0;
    const int side = white ? 0 : 8;
    for (int t = 0; t < 64; t++) {
        const int p = b->sq[t];
        if (!p || (p & 8) != side) continue;
        const int type = p & 7;
        if (type == PAWN) {
            const int tr = rank_of(t) + (white ? 1 : -1);
            if (tr == rank_of(s) && abs(file_of(t) - file_of(s)) == 1) m |= bit(t);
        } else if (attacks_from(b, type, t) & bit(s)) m |= bit(t);
    }
    return m;
}

void gen_legal(const cbv_board* cb, std::vector<cbv_move>& out)
{
    cbv_board* b = const_cast<cbv_board*>(cb); // make/unmake restores it
    std::vector<cbv_move> pseudo;
    const int k = king_square(b, b->turn);
    u64 checkers = k >= 0 ? attackers_of(b, k, !b->turn) : 0;
    if (checkers) {
        // python-chess _generate_evasions: king steps first, then captures/blocks of a single checker
        u64 sliders = 0, attacked_line = 0;
        for (int s = 0; s < 64; s++)
            if ((checkers & bit(s)) && ((b->sq[s] & 7) == BISHOP || (b->sq[s] & 7) == ROOK || (b->sq[s] & 7) == QUEEN)) sliders |= bit(s);
        u64 sl = sliders;
        while (sl) {
            const int c = msb(sl);
            sl &= ~bit(c);
            attacked_line |= line_mask(k, c) & ~bit(c);
        }
        u64 ours = 0;
        for (int s = 0; s < 64; s++)
            if (own(b, b->sq[s])) ours |= bit(s);
        u64 t = step_mask(k, KG, 8) & ~ours & ~attacked_line;
        while (t) {
            const int to = msb(t);
            t &= ~bit(to);
            pseudo.push_back(mk(k, to));
        }
        const int checker = msb(checkers);
        if (bit(checker) == checkers) {
            const u64 target = between_mask(k, checker) | checkers;
            gen_pseudo(b, pseudo, ~bit(k), target);
            if (b->ep >= 0 && !(bit(b->ep) & target)) {
                const int last_double = b->ep + (b->turn ? -8 : 8);
                if (last_double == checker) gen_ep(b, pseudo, ~0ull, ~0ull);
            }
        }
    } else gen_pseudo(b, pseudo, ~0ull, ~0ull);
    for (cbv_move m : pseudo)
        if (leaves_king_safe(b, m)) out.push_back(m);
}

bool legal(const cbv_board* b, cbv_move m)
{
    std::vector<cbv_move> mv;
    gen_legal(b, mv);
    for (cbv_move x : mv)
        if (x == m) return true;
    return false;
}

bool capture(const cbv_board* b, cbv_move m)
{
    const int to = m_to(m);
    return enemy(b, b->sq[to]) || is_ep_move(b, m);
}

const char* START_FEN = "rnbqkbnr/pppppppp/8/8/8/8/PPPPPPPP/RNBQKBNR w KQkq - 0 1";
const char PIECE_CHARS[] = ".pnbrqk";

bool parse_fen(cbv_board* b, const char* fen)
{
    cbv_board t;
    memset(t.sq, 0, sizeof(t.sq));
    std::vector<std::string> parts;
    {
        std::string cur;
        for (const char* p = fen; *p; p++) {
            if (*p == ' ') {
                if (!cur.empty()) parts.push_back(cur);
                cur.clear();
            } else cur.push_back(*p);
        }
        if (!cur.empty()) parts.push_back(cur);
    }
    if (parts.empty()) return false;
    int r = 7, f = 0;
    for (char c : parts[0]) {
        if (c == '/') {
            if (f != 8) return false;
            r--;
            f = 0;
        } else if (c >= '1' && c <= '8') f += c - '0';
        else {
            const char lc = (char)(c | 32);
            const char* q = strchr(PIECE_CHARS + 1, lc);
            if (!q || r < 0 || f > 7) return false;
            t.sq[r * 8 + f] = (int8_t)((int)(q - PIECE_CHARS) | ((c & 32) ? 8 : 0));
            f++;
        }
        if (f > 8 || r < 0) return false;
    }
    if (r != 0 || f != 8) return false;
    t.turn = 1;
    if (parts.size() > 1) {
        if (parts[1] == "w") t.turn = 1;
        else if (parts[1] == "b") t.turn = 0;
        else return false;
    }
    t.castling = 0;
    if (parts.size() > 2 && parts[2] != "-")
        for (char c : parts[2]) {
            if (c == 'K') t.castling |= WK;
            else if (c == 'Q') t.castling |= WQ;
            else if (c == 'k') t.castling |= BK;
            else if (c == 'q') t.castling |= BQ;
            else return false;
        }
    t.ep = -1;
    if (parts.size() > 3 && parts[3] != "-") {
        if (parts[3].size() != 2 || parts[3][0] < 'a' || parts[3][0] > 'h' || parts[3][1] < '1' || parts[3][1] > '8') return false;
        t.ep = (parts[3][1] - '1') * 8 + (parts[3][0] - 'a');
    }
    t.halfmove = parts.size() > 4 ? atoi(parts[4].c_str()) : 0;
    t.fullmove = parts.size() > 5 ? atoi(parts[5].c_str()) : 1;
    if (t.halfmove < 0) return false;
    if (t.fullmove < 1) t.fullmove = 1; // python-chess: max(fullmove, 1)
    memcpy(b->sq, t.sq, sizeof(t.sq));
    b->turn = t.turn;
    b->castling = t.castling;
    b->ep = t.ep;
    b->halfmove = t.halfmove;
    b->fullmove = t.fullmove;
    b->stack.clear();
    return true;
}

bool has_legal_ep(const cbv_board* b)
{
    if (b->ep < 0) return false;
    std::vector<cbv_move> mv;
    gen_legal(b, mv);
    for (cbv_move m : mv)
        if (is_ep_move(b, m)) return true;
    return false;
}

std::string emit_fen(const cbv_board* b)
{
    std::string s;
    for (int r = 7; r >= 0; r--) {
        int empty = 0;
        for (int f = 0; f < 8; f++) {
            const int p = b->sq[r * 8 + f];
            if (!p) {
                empty++;
                continue;
            }
            if (empty) s.push_back((char)('0' + empty));
            empty = 0;
            const char c = PIECE_CHARS[p & 7];
            s.push_back((p & 8) ? c : (char)(c - 32));
        }
        if (empty) s.push_back((char)('0' + empty));
        if (r) s.push_back('/');
    }
    s += b->turn ? " w " : " b ";
    const int c = clean_castling(b);
    if (!c) s += "-";
    else {
        if (c & WK) s += "K";
        if (c & WQ) s += "Q";
        if (c & BK) s += "k";
        if (c & BQ) s += "q";
    }
    s += " ";
    if (has_legal_ep(b)) {
        s.push_back((char)('a' + file_of(b->ep)));
        s.push_back((char)('1' + rank_of(b->ep)));
    } else s += "-";
    char buf[48];
    snprintf(buf, sizeof(buf), " %d %d", b->halfmove, b->fullmove);
    s += buf;
    return s;
}

u64 perft(cbv_board* b, int depth)
{
    if (depth <= 0) return 1;
    std::vector<cbv_move> mv;
    gen_legal(b, mv);
    if (depth == 1) return mv.size();
    u64 n = 0;
    for (cbv_move m : mv) {
        do_push(b, m);
        n += perft(b, depth - 1);
        do_pop(b);
    }
    return n;
}

} // namespace

extern "C" {

cbv_board* cbv_board_create(void)
{
    cbv_board* b = new cbv_board();
    parse_fen(b, START_FEN);
    return b;
}
void cbv_board_destroy(cbv_board* b) { delete b; }
void cbv_board_reset(cbv_board* b)
{
    if (b) parse_fen(b, START_FEN);
}
int cbv_board_set_fen(cbv_board* b, const char* fen) { return (b && fen && parse_fen(b, fen)) ? 0 : -1; }
int cbv_board_fen(const cbv_board* b, char* out, int cap)
{
    if (!b || !out || cap <= 0) return -1;
    const std::string s = emit_fen(b);
    snprintf(out, (size_t)cap, "%s", s.c_str());
    return (int)s.size();
}
int cbv_board_turn(const cbv_board* b) { return b ? b->turn : -1; }
void cbv_board_set_turn(cbv_board* b, int white)
{
    if (b) b->turn = white ? 1 : 0;
}
int cbv_board_piece_at(const cbv_board* b, int square) { return (b && square >= 0 && square < 64) ? b->sq[square] : 0; }
uint64_t cbv_board_occupancy(const cbv_board* b)
{
    u64 m = 0;
    if (b)
        for (int s = 0; s < 64; s++)
            if (b->sq[s]) m |= bit(s);
    return m;
}
int cbv_board_legal_moves(const cbv_board* b, cbv_move* out, int cap)
{
    if (!b) return -1;
    std::vector<cbv_move> mv;
    gen_legal(b, mv);
    for (int i = 0; i < (int)mv.size() && i < cap && out; i++) out[i] = mv[i];
    return (int)mv.size();
}
int cbv_board_is_legal(const cbv_board* b, cbv_move m) { return b && m != CBV_MOVE_NONE && legal(b, m) ? 1 : 0; }
int cbv_board_is_capture(const cbv_board* b, cbv_move m) { return b && capture(b, m) ? 1 : 0; }
int cbv_board_is_en_passant(const cbv_board* b, cbv_move m) { return b && is_ep_move(b, m) ? 1 : 0; }
int cbv_board_is_check(const cbv_board* b)
{
    if (!b) return 0;
    const int k = king_square(b, b->turn);
    return k >= 0 && attacked(b, k, !b->turn) ? 1 : 0;
}
int cbv_board_push(cbv_board* b, cbv_move m)
{
    if (!b || m == CBV_MOVE_NONE) return -1;
    do_push(b, m);
    return 0;
}
cbv_move cbv_board_pop(cbv_board* b) { return b ? do_pop(b) : (cbv_move)CBV_MOVE_NONE; }
int cbv_board_ply(const cbv_board* b) { return b ? (int)b->stack.size() : 0; }
cbv_move cbv_board_peek(const cbv_board* b) { return (b && !b->stack.empty()) ? b->stack.back().m : (cbv_move)CBV_MOVE_NONE; }
uint64_t cbv_board_perft(cbv_board* b, int depth) { return b ? perft(b, depth) : 0; }

static const char* kStatus[] = {"no_valid_change", "move_confirmed", "illegal_move", "castling_confirmed",
                                "en_passant_confirmed", "capture_confirmed", "ambiguous_capture"};
const char* cbv_game_status_name(int status) { return (status >= 0 && status < 7) ? kStatus[status] : ""; }

int cbv_game_process_occupancy(cbv_board* b, uint64_t vision, cbv_move* move_out)
{
    if (move_out) *move_out = CBV_MOVE_NONE;
    if (!b) return CBV_GAME_NO_VALID_CHANGE;
    const u64 logical = cbv_board_occupancy(b);
    const u64 vanished = logical & ~vision, appeared = vision & ~logical;
    const int nv = __builtin_popcountll(vanished), na = __builtin_popcountll(appeared);
    std::vector<cbv_move> mv;
    gen_legal(b, mv);
    auto in_legal = [&](cbv_move m) {
        for (cbv_move x : mv)
            if (x == m) return true;
        return false;
    };
    auto confirm = [&](cbv_move m, int status) {
        do_push(b, m);
        if (move_out) *move_out = m;
        return status;
    };
    if (nv == 1 && na == 1) { // normal move, queen promotion when the plain move is not legal (game_state.py:172-195)
        const int src = msb(vanished), dst = msb(appeared);
        if (in_legal(mk(src, dst))) return confirm(mk(src, dst), CBV_GAME_MOVE_CONFIRMED);
        if (in_legal(mk(src, dst, QUEEN))) return confirm(mk(src, dst, QUEEN), CBV_GAME_MOVE_CONFIRMED);
        return CBV_GAME_ILLEGAL_MOVE;
    }
    if (nv == 2 && na == 2) { // castling: the king left, a square two files away on its rank appeared (:114-137)
        for (u64 v = vanished; v;) {
            const int s = msb(v);
            v &= ~bit(s);
            if ((b->sq[s] & 7) != KING) continue;
            for (u64 a = appeared; a;) {
                const int t = msb(a);
                a &= ~bit(t);
                if (abs(file_of(t) - file_of(s)) == 2 && rank_of(t) == rank_of(s) && in_legal(mk(s, t)))
                    return confirm(mk(s, t), CBV_GAME_CASTLING_CONFIRMED);
            }
        }
    }
    if (nv == 2 && na == 1) { // en passant: attacker and victim left, the attacker appeared (:139-160)
        const int dst = msb(appeared);
        for (u64 v = vanished; v;) {
            const int s = msb(v);
            v &= ~bit(s);
            if ((b->sq[s] & 7) != PAWN) continue;
            const cbv_move m = mk(s, dst);
            if (in_legal(m) && is_ep_move(b, m)) return confirm(m, CBV_GAME_EN_PASSANT_CONFIRMED);
        }
    }
    if (nv == 1 && na == 0) { // capture: the attacker left and now stands on a square that was occupied (:162-183)
        const int src = msb(vanished);
        int n = 0;
        cbv_move cand = CBV_MOVE_NONE;
        for (cbv_move m : mv)
            if (m_from(m) == src && capture(b, m) && (vision & bit(m_to(m)))) {
                if (n == 0) cand = m;
                n++;
            }
        if (n == 1) return confirm(cand, CBV_GAME_CAPTURE_CONFIRMED);
        if (n > 1) return CBV_GAME_AMBIGUOUS_CAPTURE;
    }
    return CBV_GAME_NO_VALID_CHANGE;
}

int cbv_game_infer_move(const cbv_board* b, uint64_t vision, cbv_move* move_out)
{
    if (move_out) *move_out = CBV_MOVE_NONE;
    if (!b) return 0;
    const u64 logical = cbv_board_occupancy(b);
    const u64 missing = logical & ~vision, extra = vision & ~logical;
    std::vector<cbv_move> mv, cand;
    gen_legal(b, mv);
    auto in_legal = [&](cbv_move m) {
        for (cbv_move x : mv)
            if (x == m) return true;
        return false;
    };
    auto add = [&](cbv_move m) {
        for (cbv_move x : cand)
            if (x == m) return;
        cand.push_back(m);
    };
    // 1. origin vanished, destination appeared (queen promotion when the plain move is not legal)   game_session.py:235-248
    for (u64 o = missing; o;) {
        const int s = msb(o);
        o &= ~bit(s);
        for (u64 e = extra; e;) {
            const int t = msb(e);
            e &= ~bit(t);
            if (in_legal(mk(s, t))) add(mk(s, t));
            else if (in_legal(mk(s, t, QUEEN))) add(mk(s, t, QUEEN));
        }
    }
    // 2. captures from a vanished origin onto a square vision still sees occupied   game_session.py:250-257
    for (cbv_move m : mv)
        if ((missing & bit(m_from(m))) && capture(b, m) && (vision & bit(m_to(m)))) add(m);
    if (cand.size() == 1 && move_out) *move_out = cand[0];
    return (int)cand.size();
}

uint64_t cbv_roi_bits_to_squares(uint64_t roi_bits)
{
    u64 out = 0;
    for (int i = 0; i < 64; i++)
        if ((roi_bits >> i) & 1) out |= bit((7 - (i >> 3)) * 8 + (i & 7));
    return out;
}

} // extern "C"
