"""Host file formats of the trainer: entity-grouped and per-record TFRecord in, photon-ml Avro model and score files out.
`tfrecord`, `grouped_reader`, `avro` state the formats in Python; `native_reader` binds libgdmix_io.so, which is what runs."""
