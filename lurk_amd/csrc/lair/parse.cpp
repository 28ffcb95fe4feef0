// Text front end for Lair functions: the surface syntax of the reference's `func!` macro
// (/root/reference/src/lair/macros.rs) parsed at run time, so that host programs and tests can
// state functions exactly as the reference's sources do.
#include <cctype>
#include <cstdlib>

#include "lair.h"

namespace lair {

namespace {

struct Tok {
    enum Kind { Ident, Num, Punct, Str, End } kind = End;
    std::string text;
    int64_t num = 0;
    size_t pos = 0;
};

struct Lexer {
    const std::string& s;
    size_t i = 0;
    explicit Lexer(const std::string& src) : s(src) {}

    [[noreturn]] void fail(const std::string& msg, size_t pos) const {
        size_t line = 1;
        for (size_t k = 0; k < pos && k < s.size(); k++)
            if (s[k] == '\n') line++;
        throw ParseError("lair parse error at line " + std::to_string(line) + ": " + msg);
    }

    void skip_ws() {
        for (;;) {
            while (i < s.size() && isspace((unsigned char)s[i])) i++;
            if (i + 1 < s.size() && s[i] == '/' && s[i + 1] == '/') {
                while (i < s.size() && s[i] != '\n') i++;
                continue;
            }
            break;
        }
    }

    Tok next() {
        skip_ws();
        Tok t;
        t.pos = i;
        if (i >= s.size()) return t;
        char c = s[i];
        if (isalpha((unsigned char)c) || c == '_') {
            size_t j = i;
            while (j < s.size() && (isalnum((unsigned char)s[j]) || s[j] == '_' || (s[j] == ':' && j + 1 < s.size() && s[j + 1] == ':'))) {
                if (s[j] == ':') j += 2;
                else j++;
            }
            if (j < s.size() && s[j] == '!' && !(j + 1 < s.size() && s[j + 1] == '=')) j++;  // macro-style names
            t.kind = Tok::Ident;
            t.text = s.substr(i, j - i);
            i = j;
            return t;
        }
        if (isdigit((unsigned char)c) || (c == '-' && i + 1 < s.size() && isdigit((unsigned char)s[i + 1]))) {
            size_t j = i + 1;
            while (j < s.size() && (isalnum((unsigned char)s[j]) || s[j] == '_')) j++;
            std::string lit;
            for (size_t k = i; k < j; k++)
                if (s[k] != '_') lit.push_back(s[k]);
            t.kind = Tok::Num;
            t.num = strtoll(lit.c_str(), nullptr, 0);
            t.text = lit;
            i = j;
            return t;
        }
        if (c == '"') {
            size_t j = i + 1;
            while (j < s.size() && s[j] != '"') j++;
            if (j >= s.size()) fail("unterminated string", i);
            t.kind = Tok::Str;
            t.text = s.substr(i + 1, j - i - 1);
            i = j + 1;
            return t;
        }
        t.kind = Tok::Punct;
        if (c == '=' && i + 1 < s.size() && s[i + 1] == '>') {
            t.text = "=>";
            i += 2;
            return t;
        }
        t.text = std::string(1, c);
        i++;
        return t;
    }
};

using Env = std::map<std::string, int>;  // variable name -> size in scope

struct Parser {
    Lexer lx;
    Tok cur;
    const std::map<std::string, uint32_t>* symbols;
    Parser(const std::string& src, const std::map<std::string, uint32_t>* sy) : lx(src), symbols(sy) { cur = lx.next(); }

    [[noreturn]] void fail(const std::string& m) { lx.fail(m + " (near '" + cur.text + "')", cur.pos); }
    void advance() { cur = lx.next(); }
    bool is_punct(const char* p) const { return cur.kind == Tok::Punct && cur.text == p; }
    bool is_ident(const char* p) const { return cur.kind == Tok::Ident && cur.text == p; }
    void expect_punct(const char* p) {
        if (!is_punct(p)) fail(std::string("expected '") + p + "'");
        advance();
    }
    std::string expect_ident() {
        if (cur.kind != Tok::Ident) fail("expected identifier");
        std::string t = cur.text;
        advance();
        return t;
    }
    int64_t expect_num() {
        if (cur.kind != Tok::Num) fail("expected number");
        int64_t v = cur.num;
        advance();
        return v;
    }
    uint32_t constant_value() {
        if (cur.kind == Tok::Num) return field_from_i64(expect_num());
        if (cur.kind == Tok::Ident) {
            if (symbols) {
                auto it = symbols->find(cur.text);
                if (it != symbols->end()) {
                    advance();
                    return it->second % P;
                }
            }
            fail("unknown constant symbol");
        }
        fail("expected constant");
    }

    Var use(const Env& env, const std::string& n) {
        auto it = env.find(n);
        if (it == env.end()) fail("variable " + n + " is unbound");
        return Var{n, it->second};
    }

    // `[size]` after ':'
    int parse_size_annot() {
        expect_punct("[");
        int64_t v = expect_num();
        expect_punct("]");
        return (int)v;
    }

    std::vector<Var> parse_args(const Env& env) {  // ident, ident, ... up to ')'
        std::vector<Var> v;
        while (!is_punct(")")) {
            v.push_back(use(env, expect_ident()));
            if (is_punct(",")) advance();
            else break;
        }
        expect_punct(")");
        return v;
    }

    static int total(const std::vector<Var>& v) {
        int s = 0;
        for (auto& x : v) s += x.size;
        return s;
    }

    void parse_let(Env& env, std::vector<OpE>& ops) {
        // targets
        std::vector<Var> tg;
        bool paren = false;
        bool sized = false;
        if (is_punct("(")) {
            paren = true;
            advance();
            while (!is_punct(")")) {
                Var v{expect_ident(), 1};
                if (is_punct(":")) {
                    advance();
                    v.size = parse_size_annot();
                }
                tg.push_back(v);
                if (is_punct(",")) advance();
                else break;
            }
            expect_punct(")");
        } else {
            Var v{expect_ident(), 1};
            if (is_punct(":")) {
                advance();
                v.size = parse_size_annot();
                sized = true;
            }
            tg.push_back(v);
        }
        expect_punct("=");
        OpE op;
        auto bind_all = [&]() {
            for (auto& v : tg) env[v.name] = v.size;
        };
        if (cur.kind == Tok::Num || (cur.kind == Tok::Ident && !paren && !is_call_like())) {
            // let x = <literal or symbolic constant>;
            if (paren || tg.size() != 1) fail("constant needs a single target");
            op.kind = OpEKind::Const;
            op.consts = {constant_value()};
            tg[0].size = 1;
            op.out = tg;
        } else if (is_punct("[")) {
            advance();
            List arr;
            if (!is_punct("]")) {
                uint32_t first = constant_value();
                if (is_punct(";")) {
                    advance();
                    int64_t n = expect_num();
                    arr.assign((size_t)n, first);
                } else {
                    arr.push_back(first);
                    while (is_punct(",")) {
                        advance();
                        if (is_punct("]")) break;
                        arr.push_back(constant_value());
                    }
                }
            }
            expect_punct("]");
            op.kind = OpEKind::Array;
            op.consts = arr;
            tg[0].size = (int)arr.size();
            op.out = tg;
        } else if (is_punct("(")) {
            // slice: let x: [n] = (a, b);   let (a, b) = (x, y);
            advance();
            op.kind = OpEKind::Slice;
            op.in = parse_args(env);
            op.out = tg;
        } else if (cur.kind == Tok::Ident) {
            std::string fn = expect_ident();
            if (paren && !is_punct("(")) {
                // let (a, b) = x;
                op.kind = OpEKind::Slice;
                op.in = {use(env, fn)};
                op.out = tg;
            } else {
                expect_punct("(");
                auto binop = [&](OpEKind k) {
                    op.kind = k;
                    op.in = parse_args(env);
                    if (op.in.size() != 2) fail(fn + " takes two operands");
                    tg[0].size = (k == OpEKind::Eq) ? 1 : op.in[0].size;
                    op.out = tg;
                };
                if (fn == "add") binop(OpEKind::Add);
                else if (fn == "sub") binop(OpEKind::Sub);
                else if (fn == "mul") binop(OpEKind::Mul);
                else if (fn == "div") binop(OpEKind::Div);
                else if (fn == "eq") binop(OpEKind::Eq);
                else if (fn == "inv" || fn == "not") {
                    op.kind = fn == "inv" ? OpEKind::Inv : OpEKind::Not;
                    op.in = parse_args(env);
                    if (op.in.size() != 1) fail(fn + " takes one operand");
                    tg[0].size = fn == "inv" ? op.in[0].size : 1;
                    op.out = tg;
                } else if (fn == "store") {
                    op.kind = OpEKind::Store;
                    op.in = parse_args(env);
                    tg[0].size = 1;
                    op.out = tg;
                } else if (fn == "load") {
                    op.kind = OpEKind::Load;
                    op.in = parse_args(env);
                    if (op.in.size() != 1) fail("load takes one pointer");
                    op.out = tg;
                } else if (fn == "call" || fn == "extern_call" || fn == "preimg") {
                    op.kind = fn == "call" ? OpEKind::Call : fn == "preimg" ? OpEKind::PreImg : OpEKind::ExternCall;
                    op.name = expect_ident();
                    if (is_punct(",")) advance();
                    op.in = parse_args(env);
                    op.out = tg;
                } else {
                    fail("unknown operation " + fn);
                }
            }
        } else {
            fail("bad right-hand side");
        }
        (void)sized;
        ops.push_back(op);
        // the macro rebinds targets with the sizes just fixed
        tg = ops.back().out;
        for (auto& v : tg) env[v.name] = v.size;
        (void)bind_all;
        expect_punct(";");
    }

    bool is_call_like() {
        // an identifier followed by '(' is an operation, otherwise a symbolic constant
        Lexer save = lx;
        Tok t = lx.next();
        lx.i = save.i;
        return t.kind == Tok::Punct && t.text == "(";
    }

    // parses statements up to (and including) the closing '}' of the block; returns the block
    std::shared_ptr<BlockE> parse_block_body(Env env, CaseType* constrained = nullptr) {
        auto blk = std::make_shared<BlockE>();
        if (constrained) *constrained = CaseType::Constrained;
        if (is_punct("#")) {
            advance();
            expect_punct("[");
            std::string a = expect_ident();
            if (a != "unconstrained") fail("unknown attribute");
            expect_punct("]");
            if (constrained) *constrained = CaseType::Unconstrained;
        }
        for (;;) {
            if (is_ident("let")) {
                advance();
                parse_let(env, blk->ops);
            } else if (is_ident("return")) {
                advance();
                blk->ctrl.kind = CtrlEKind::Return;
                if (is_punct("(")) {
                    advance();
                    blk->ctrl.ret = parse_args(env);
                } else {
                    blk->ctrl.ret = {use(env, expect_ident())};
                }
                if (is_punct(";")) advance();
                expect_punct("}");
                return blk;
            } else if (is_ident("if")) {
                advance();
                bool negate = false;
                if (is_punct("!")) {
                    negate = true;
                    advance();
                }
                Var x = use(env, expect_ident());
                expect_punct("{");
                auto inner = parse_block_body(env);
                auto rest = parse_block_body(env);
                blk->ctrl.kind = CtrlEKind::If;
                blk->ctrl.var = x;
                blk->ctrl.t = negate ? rest : inner;
                blk->ctrl.f = negate ? inner : rest;
                return blk;
            } else if (is_ident("match")) {
                advance();
                Var x = use(env, expect_ident());
                expect_punct("{");
                bool many = false;
                std::vector<CaseE> branches;
                while (!is_punct("}")) {
                    std::vector<List> pats;
                    for (;;) {
                        if (is_punct("[")) {
                            many = true;
                            advance();
                            List arr;
                            while (!is_punct("]")) {
                                arr.push_back(constant_value());
                                if (is_punct(",")) advance();
                            }
                            expect_punct("]");
                            pats.push_back(arr);
                        } else {
                            pats.push_back({constant_value()});
                        }
                        if (is_punct(",")) {
                            advance();
                            continue;
                        }
                        break;
                    }
                    expect_punct("=>");
                    expect_punct("{");
                    CaseType ct;
                    auto b = parse_block_body(env, &ct);
                    if (is_punct(",")) advance();
                    if (many) {
                        for (auto& pat : pats) branches.push_back(CaseE{pat, b, ct});
                    } else {
                        List keys;
                        for (auto& pat : pats) keys.push_back(pat[0]);
                        branches.push_back(CaseE{keys, b, ct});
                    }
                }
                expect_punct("}");
                blk->ctrl.kind = many ? CtrlEKind::MatchMany : CtrlEKind::Match;
                blk->ctrl.var = x;
                blk->ctrl.branches = branches;
                if (is_punct(";")) {
                    advance();
                    CaseType ct;
                    blk->ctrl.def = parse_block_body(env, &ct);
                    blk->ctrl.def_constrained = ct;
                } else {
                    expect_punct("}");
                }
                return blk;
            } else if (cur.kind == Tok::Ident) {
                std::string fn = expect_ident();
                OpE op;
                if (fn == "breakpoint") {
                    op.kind = OpEKind::Breakpoint;
                } else {
                    expect_punct("(");
                    if (fn == "assert_eq!" || fn == "assert_ne!" || fn == "contains!") {
                        op.kind = fn == "assert_eq!" ? OpEKind::AssertEq : fn == "assert_ne!" ? OpEKind::AssertNe : OpEKind::Contains;
                        op.in = parse_args(env);
                        if (op.in.size() != 2) fail(fn + " takes two operands");
                    } else if (fn == "range_u8!") {
                        op.kind = OpEKind::RangeU8;
                        op.in = parse_args(env);
                    } else if (fn == "emit") {
                        op.kind = OpEKind::Emit;
                        op.in = parse_args(env);
                    } else if (fn == "debug!") {
                        op.kind = OpEKind::Debug;
                        if (cur.kind != Tok::Str) fail("debug! takes a string");
                        op.name = cur.text;
                        advance();
                        expect_punct(")");
                    } else {
                        fail("unknown statement " + fn);
                    }
                }
                blk->ops.push_back(op);
                expect_punct(";");
            } else {
                fail("unexpected token in block");
            }
        }
    }

    FuncE parse_func() {
        FuncE f;
        for (;;) {
            if (is_ident("partial")) {
                f.partial = true;
                advance();
            } else if (is_ident("invertible")) {
                f.invertible = true;
                advance();
            } else {
                break;
            }
        }
        if (!is_ident("fn")) fail("expected 'fn'");
        advance();
        f.name = expect_ident();
        expect_punct("(");
        Env env;
        while (!is_punct(")")) {
            Var v{expect_ident(), 1};
            if (is_punct(":")) {
                advance();
                v.size = parse_size_annot();
            }
            f.input_params.push_back(v);
            env[v.name] = v.size;
            if (is_punct(",")) advance();
        }
        expect_punct(")");
        expect_punct(":");
        f.output_size = parse_size_annot();
        expect_punct("{");
        f.body = *parse_block_body(env);
        return f;
    }
};

}  // namespace

std::vector<FuncE> parse_funcs(const std::string& src, const std::map<std::string, uint32_t>* symbols) {
    Parser p(src, symbols);
    std::vector<FuncE> out;
    while (p.cur.kind != Tok::End) out.push_back(p.parse_func());
    return out;
}

}  // namespace lair
