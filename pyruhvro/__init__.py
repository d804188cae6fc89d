"""Drop-in module name: ``import pyruhvro`` resolves to the MI355X-native engine.

Same five functions as the reference's PyO3 module (src/lib.rs:150-158)."""
from pyruhvro_amd import (  # noqa: F401
    deserialize_array,
    deserialize_array_threaded,
    deserialize_array_threaded_spawn,
    serialize_record_batch,
    serialize_record_batch_spawn,
)
