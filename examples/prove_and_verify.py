"""End to end on one MI355X: a Lair program is executed, proved in shards on the GPU, stored in the reference's serialised form
and verified from the stored bytes on the host alone.

    python examples/prove_and_verify.py [n]

Mirrors the reference's flow (/root/reference/benches/fib.rs:88-133, /root/reference/src/core/cli/repl.rs:164-207):
`toplevel.execute` -> `machine.setup` -> `machine.prove` -> `bincode::serialize(CryptoProof)` -> `machine.verify`."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))  # run from anywhere: the package lives beside examples/

import lurk_amd
from lurk_amd import lair, proofs, prover

SOURCE = """
partial fn fib(n): [1] {
    let one = 1;
    match n {
        0 => {
            let zero = 0;
            return zero
        }
        1 => {
            return one
        }
    };
    let n_1 = sub(n, one);
    let a = call(fib, n_1);
    let n_2 = sub(n_1, one);
    let b = call(fib, n_2);
    let res = add(a, b);
    return res
}
"""


def main(n: int = 3000):
    top = lair.Toplevel(SOURCE)
    record = lair.QueryRecord(top)
    t0 = time.perf_counter()
    out = top.execute_by_name("fib", [n], record)
    public_values = record.expect_public_values()  # [n, fib(n) mod p, depth as 4 bytes]
    t_exec = time.perf_counter() - t0
    with lurk_amd.Context(0) as ctx:
        machine = prover.Machine(ctx, top, "fib", len(public_values))
        machine.setup()
        t0 = time.perf_counter()
        shard_proofs = machine.prove(record, lair.ShardingConfig(1 << 10), num_queries=100, pow_bits=16)
        t_prove = time.perf_counter() - t0
        data = proofs.CryptoProof.from_machine_proof(machine, shard_proofs).to_bytes()
        t0 = time.perf_counter()
        proofs.verify_crypto_proof(machine, data, public_values, num_queries=100, pow_bits=16)  # host only: no kernel runs here
        t_verify = time.perf_counter() - t0
        tampered = bytearray(data)
        tampered[len(data) // 2] ^= 1
        try:
            proofs.verify_crypto_proof(machine, bytes(tampered), public_values, num_queries=100, pow_bits=16)
            raise SystemExit("a tampered proof was accepted")
        except prover.VerificationError as e:
            rejected = str(e)
        machine.close()
    print(f"fib({n}) mod p = {out[0]}: {record.num_func_queries(0)} queries executed in {t_exec * 1e3:.1f} ms, {len(shard_proofs)} shard proofs in "
          f"{t_prove * 1e3:.1f} ms, {len(data)} bytes, verified from the bytes in {t_verify * 1e3:.1f} ms; a flipped bit: {rejected}")
    return out[0]


if __name__ == "__main__":
    main(int(sys.argv[1]) if len(sys.argv) > 1 else 3000)
