"""Synthetic Avro inputs for tests and bench.py (test/bench infrastructure, not product code).

Faker / fastavro / apache-avro are not available in this image, so the wire
bytes are produced by a small spec-following encoder (``encoder``) for the
parity tests and by a C generator (``fastgen``) for the 1M/10M-record
BASELINE.json configurations.
"""
