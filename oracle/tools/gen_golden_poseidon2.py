#!/usr/bin/env python3
"""Golden vectors for Poseidon2-t8 arity-7 `hash`, produced by RUNNING the reference.

The reference's only importable code on the proving path is the Python generator of its
on-chain hasher, /root/reference/poseidon2-solidity/generate_t8.py (+ utils.py), whose output
(Yul assembly inside Poseidon2T8Assembly.sol) the reference asserts equal to its off-circuit
`hash` (/root/reference/crates/integration-tests/src/poseidon2.rs:32-50).  This script imports
the generator, takes the Yul text it emits and executes it with the small Yul evaluator below
(EVM word semantics: add wraps mod 2^256, addmod/mulmod as the EVM defines them), then writes
inputs and outputs to tests/golden/poseidon2_t8.json.  Only the vectors are committed; nothing
of the reference travels.  Run in the build container:  python3 oracle/tools/gen_golden_poseidon2.py
"""
import json, os, random, re, sys

sys.dont_write_bytecode = True
sys.path.insert(0, "/root/reference/poseidon2-solidity")
import generate_t8 as g  # noqa: E402

W = 1 << 256
TOK = re.compile(r"\s*(?:(//[^\n]*)|(0x[0-9a-fA-F]+|\d+)|([A-Za-z_][A-Za-z_0-9]*)|(:=)|([(){},]))")


def tokenize(src):
    out, pos = [], 0
    while pos < len(src):
        m = TOK.match(src, pos)
        if not m:
            if src[pos:].strip() == "":
                break
            raise SyntaxError(src[pos:pos + 40])
        pos = m.end()
        if m.group(1):
            continue
        out.append(m.group(2) or m.group(3) or m.group(4) or m.group(5))
    return out


class Yul:
    def __init__(self, toks):
        self.t, self.i, self.funcs, self.mem, self.ret = toks, 0, {}, {}, None

    def peek(self): return self.t[self.i] if self.i < len(self.t) else None
    def eat(self, x=None):
        v = self.t[self.i]; self.i += 1
        if x is not None and v != x: raise SyntaxError(f"expected {x} got {v}")
        return v

    # ---- parse
    def block(self):
        self.eat("{"); stmts = []
        while self.peek() != "}": stmts.append(self.stmt())
        self.eat("}"); return ("block", stmts)

    def stmt(self):
        p = self.peek()
        if p == "{": return self.block()
        if p == "function":
            self.eat(); name = self.eat(); self.eat("("); params = []
            while self.peek() != ")":
                params.append(self.eat())
                if self.peek() == ",": self.eat()
            self.eat(")"); body = self.block(); self.funcs[name] = (params, body); return ("nop",)
        if p == "let":
            self.eat(); name = self.eat()
            if self.peek() == ":=": self.eat(); return ("let", name, self.expr())
            return ("let", name, ("num", 0))
        if self.t[self.i + 1] == ":=":
            name = self.eat(); self.eat(":="); return ("set", name, self.expr())
        return ("expr", self.expr())

    def expr(self):
        v = self.eat()
        if v[0].isdigit(): return ("num", int(v, 0))
        if self.peek() == "(":
            self.eat("("); args = []
            while self.peek() != ")":
                args.append(self.expr())
                if self.peek() == ",": self.eat()
            self.eat(")"); return ("call", v, args)
        return ("var", v)

    # ---- eval
    def ev(self, e, env):
        k = e[0]
        if k == "num": return e[1]
        if k == "var":
            for sc in reversed(env):
                if e[1] in sc: return sc[e[1]]
            raise NameError(e[1])
        name, args = e[1], [self.ev(a, env) for a in e[2]]
        if name == "add": return (args[0] + args[1]) % W
        if name == "addmod": return (args[0] + args[1]) % args[2]
        if name == "mulmod": return (args[0] * args[1]) % args[2]
        if name == "mload": return self.mem.get(args[0], 0)
        if name == "mstore": self.mem[args[0]] = args[1] % W; return None
        if name == "return": self.ret = self.mem.get(args[0], 0); return None
        params, body = self.funcs[name]
        self.run(body, [dict(zip(params, args))]); return None

    def run(self, node, env):
        k = node[0]
        if k == "block":
            env = env + [{}]
            for s in node[1]:
                self.run(s, env)
                if self.ret is not None: return
        elif k == "let": env[-1][node[1]] = self.ev(node[2], env)
        elif k == "set":
            v = self.ev(node[2], env)
            for sc in reversed(env):
                if node[1] in sc: sc[node[1]] = v; break
            else: raise NameError(node[1])
        elif k == "expr": self.ev(node[1], env)


def reference_hash(inputs):
    sol = g.generate_code(g.init, g.full_round, g.partial_round, g.T, g.ROUNDS_F, g.ROUNDS_P, g.FUNCTION_COMMENT)
    body = sol[sol.index("assembly {") + len("assembly "):]
    y = Yul(tokenize(body))
    tree = y.block()
    for i, v in enumerate(inputs):            # uint256[7] memory argument at 0x80 (utils.ARG)
        y.mem[0x80 + 0x20 * i] = v
    y.run(tree, [{}])
    assert y.ret is not None
    return y.ret


def main():
    rnd = random.Random(20251003)
    cases = [list(range(1, 8)), [0] * 7, [g.F - 1] * 7, [1, 0, 0, 0, 0, 0, 0]]
    cases += [[rnd.randrange(g.F) for _ in range(7)] for _ in range(12)]
    vec = [{"inputs": ["0x%064x" % x for x in c], "hash": "0x%064x" % reference_hash(c)} for c in cases]
    out = os.path.join(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))), "tests", "golden", "poseidon2_t8.json")
    with open(out, "w") as f:
        json.dump({"source": "executed Yul emitted by /root/reference/poseidon2-solidity/generate_t8.py",
                   "field_modulus": "0x%064x" % g.F, "vectors": vec}, f, indent=1)
    print("wrote", out, vec[0]["hash"])


if __name__ == "__main__":
    main()
