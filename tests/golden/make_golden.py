"""Generates tests/golden/reference_vectors.json.

The hex datums are the golden vectors the reference's own tests carry
(ruhvro/src/lib.rs:165-167, ruhvro/src/deserialize.rs:244,303 -- produced by
fastavro + Faker upstream).  The reference only asserts shapes on them; the
expected VALUES stored here were obtained by decoding the bytes by hand from
the Avro 1.11 wire rules and are written
out literally below, so the fixture pins the oracle rather than the other way
round.  The script then double-checks the literals against oracle/py_walker
and refuses to write the file on any disagreement.

The reference is Rust (no cargo/rustc in this image), so no reference binary
can generate vectors here; run:  python tests/golden/make_golden.py
"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)

from avrogen.schemas import SCHEMAS  # noqa: E402

VECTORS = [
    {
        "name": "lib_rs_165_avro_datum", "schema": "kat_user", "source": "ruhvro/src/lib.rs:165",
        "hex": "0000062e74686f6d61736b6172656e406578616d706c652e6e657422616c6f7765406578616d706c652e6f72672664617669643738406578616d706c652e636f6d0000060a636865636b203030312d3233372d3438302d353133341065766964656e6365262b312d3732352d3336362d39323133783730300a6d616a6f722428393734293537302d3032313178333534350002020a656d61696c00020a7374616666",
        "consumed": 157,
        "expected": {"name": None, "age": None,
                     "emails": ["thomaskaren@example.net", "alowe@example.org", "david78@example.com"],
                     "address": None,
                     "phone_numbers": [["check", "001-237-480-5134"], ["evidence", "+1-725-366-9213x700"],
                                       ["major", "(974)570-0211x3545"]],
                     "preferences": {"contact_method": "email", "newsletter": False},
                     "status": 5, "status_type_id": 1},
    },
    {
        "name": "lib_rs_166_avro_datum2", "schema": "kat_user", "source": "ruhvro/src/lib.rs:166",
        "hex": "0218416d616e646120456c6c6973023804246e6361736579406578616d706c652e636f6d307374657761727474796c6572406578616d706c652e6e6574000230393532323120436861726c657320547261666669637761791c5a616368617279626f726f7567680a303433343300000202",
        "consumed": 113,
        "expected": {"name": "Amanda Ellis", "age": 28,
                     "emails": ["ncasey@example.com", "stewarttyler@example.net"],
                     "address": {"street": "95221 Charles Trafficway", "city": "Zacharyborough", "zipcode": "04343"},
                     "phone_numbers": [], "preferences": None, "status": 1, "status_type_id": 1},
    },
    {
        "name": "lib_rs_167_avro_datum3", "schema": "kat_user", "source": "ruhvro/src/lib.rs:167",
        "hex": "021a417564726579204261726e65730000000408726963682628373136293338322d363937327837383437320e746f6e69676874243536302e3736352e3230363378373831313500020000000866726565",
        "consumed": 81,
        "expected": {"name": "Audrey Barnes", "age": None, "emails": [], "address": None,
                     "phone_numbers": [["rich", "(716)382-6972x78472"], ["tonight", "560.765.2063x78115"]],
                     "preferences": {"contact_method": None, "newsletter": False},
                     "status": "free", "status_type_id": 0},
    },
    {
        "name": "deserialize_rs_244", "schema": "kat_userdata", "source": "ruhvro/src/deserialize.rs:244",
        "hex": "4834346437643065662d613264662d343833652d393261312d313532333830366164656334380a4c696e64610857617265022c6c696e646173636f7474406578616d706c652e6e6574062628323636293734302d31323737783031313432283030312d3935392d3839342d36353030783739392a3030312d3339362d3831392d363830307830303139000006044d72100866696e640e10617070726f6163680c00c0f691c7c35f",
        "consumed": 167,
        "expected": {"userId": "44d7d0ef-a2df-483e-92a1-1523806adec4", "age": 28,
                     "fullName": {"firstName": "Linda", "lastName": "Ware"},
                     "email": "lindascott@example.net",
                     "phoneNumbers": ["(266)740-1277x01142", "001-959-894-6500x799", "001-396-819-6800x0019"],
                     "isPremiumMember": False,
                     "favoriteItems": [["Mr", 8], ["find", 7], ["approach", 6]],
                     "registrationDate_ms": 1641154756000},
    },
    {
        "name": "deserialize_rs_303", "schema": "kat_addresses", "source": "ruhvro/src/deserialize.rs:303",
        "hex": "084a6f686e06446f653c041431323320456c6d20537412536f6d6577686572650a313233343514343536204f616b20537410416e7977686572650a36373839300002286a6f686e2e646f65406578616d706c652e636f6d",
        "consumed": 87,
        "expected": {"firstName": "John", "lastName": "Doe", "age": 30,
                     "addresses": [{"street": "123 Elm St", "city": "Somewhere", "zipCode": "12345"},
                                   {"street": "456 Oak St", "city": "Anywhere", "zipCode": "67890"}],
                     "email": "john.doe@example.com"},
    },
]


def normalise(row: dict, batch, i: int) -> dict:
    """pyarrow row -> the literal form used in VECTORS."""
    out = {}
    for k, v in row.items():
        if isinstance(v, list) and v and isinstance(v[0], tuple):
            v = [list(t) for t in v]
        if k == "registrationDate":
            out["registrationDate_ms"] = batch.column("registrationDate").cast("int64")[i].as_py()
            continue
        out[k] = v
    if "status" in row:
        out["status_type_id"] = batch.column("status").type_codes[i].as_py()
    return out


def main():
    from oracle import py_walker
    for v in VECTORS:
        rb = py_walker.decode([bytes.fromhex(v["hex"])], SCHEMAS[v["schema"]])
        got = normalise(rb.to_pylist()[0], rb, 0)
        if got != v["expected"]:
            raise SystemExit(f"{v['name']}: oracle disagrees with the hand-decoded literal:\n{got}\n{v['expected']}")
        # the oracle must stop exactly where the hand decode stops (trailing bytes are ignored)
        short = py_walker.decode([bytes.fromhex(v["hex"])[:v["consumed"]]], SCHEMAS[v["schema"]])
        assert short.equals(rb)
        try:
            py_walker.decode([bytes.fromhex(v["hex"])[:v["consumed"] - 1]], SCHEMAS[v["schema"]])
            raise SystemExit(f"{v['name']}: consumed length is not minimal")
        except py_walker.DecodeError:
            pass
    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "reference_vectors.json")
    with open(path, "w") as f:
        json.dump({"schemas": {v["schema"]: json.loads(SCHEMAS[v["schema"]]) for v in VECTORS}, "vectors": VECTORS},
                  f, indent=1)
    print("wrote", path)


if __name__ == "__main__":
    main()
