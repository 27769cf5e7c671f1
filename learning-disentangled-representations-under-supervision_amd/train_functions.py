"""The alternating wake-sleep training loop and its two validation loops, behind the reference's entry points
(`train_functions.py`: generate_samples_for_disc_training :7, train_complete_model :19, enc_gen_validation :152,
disc_validation :224 — same names, arguments and call order of the model's closures).

What differs from the reference is where the host waits for the GPU: the reference copies eleven loss scalars to the
host after EVERY step (:59-63, :92-97); here the closures return device scalars, the loop keeps them on the device and
reads them only at the reference's own logging points (every 5th discriminator step :65, every 50th encoder/generator
step :99) and in the validation summaries. At those points, where the host waits anyway, the persistent-LSTM timeout
flag is checked (hip/ops.py raise_on_lstm_timeout): a launch that could not get its workgroups resident together must
never be trained on silently. `summary_writer` is any object with add_scalar(tag, value, step) or None; tag names are
the reference's (:124-150, :212-219, :257-260). Not rebuilt: the legacy single-model trainers (:265-559).
"""
import os

import torch as t

from .hip import ops as _ops

GEN_SAMPLES_PER_CALL = 10      # train_functions.py:39: sentences per sampler call (beam memory grows with it)
ENC_GEN_VALID_BATCH = 500      # train_functions.py:165
DISC_LOG_EVERY, DISC_VALID_EVERY = 5, 40              # :65, :69
ENC_GEN_LOG_EVERY, ENC_GEN_VALID_EVERY = 50, 100      # :99, :102
CHECKPOINT_EVERY = 5000                               # :107

_DISC_TAGS = ("total_loss", "labeled_cross_entropy", "generated_total_loss", "generated_cross_entropy",
              "generated_emperical_shannon")


def generate_samples_for_disc_training(cgn_model, data_handler, config, use_cuda, number_of_gen_samples,
                                       gen_samples_batch_size):
    """train_functions.py:7-17: number_of_gen_samples sentences from the generator, drawn gen_samples_batch_size at a
    time (integer division: the reference is Python 2), with the codes c they were conditioned on, stacked [n, C].
    A model without a sampler (NotImplementedError) is served by data_handler.synthetic_generated_samples."""
    sentences, codes = [], []
    calls = number_of_gen_samples // gen_samples_batch_size
    try:
        for _ in range(calls):
            s, c = cgn_model.sample_generator_for_learning(data_handler, config, use_cuda, sample=gen_samples_batch_size)
            sentences += list(s)
            codes += list(c)
    except NotImplementedError:
        return data_handler.synthetic_generated_samples(number_of_gen_samples)
    return sentences, t.stack(codes, dim=0)


class _Scalars:
    """Routes named scalars to print + the summary writer; .item() (a device sync) happens only here."""

    def __init__(self, writer, verbose):
        self.writer, self.verbose = writer, verbose

    def emit(self, section, step, pairs):
        _ops.raise_on_lstm_timeout()  # the host is about to wait for the device anyway
        vals = [(k, float(v)) for k, v in pairs]
        if self.verbose:
            print("[%s] step %d  " % (section, step) + "  ".join("%s %.5f" % kv for kv in vals))
        if self.writer is not None:
            for k, v in vals:
                self.writer.add_scalar("%s/%s" % (section, k), v, step)
        return dict(vals)


def _mean_on_device(rows):
    """rows: list of tuples of 1-element device tensors -> tuple of their column means (still on the device)."""
    return tuple(t.stack([r[j].reshape(()) for r in rows]).mean() for j in range(len(rows[0])))


def enc_gen_validation(cgn_model, data_handler, use_cuda, step, valid_enc_gen, summary_writer, batch_size=None,
                       verbose=True):
    """train_functions.py:152-222: forward-only pass over the validation lines in batches of 500, c inferred by the
    discriminator per batch (:176), calc_z_loss=False (:184); returns the means {name: value} it logs."""
    bs = ENC_GEN_VALID_BATCH if batch_size is None else int(batch_size)
    loader = data_handler.gen_batch_loader
    n_iter = loader.val_lines // bs
    cgn_model.prep_enc_gen_validation()
    rows, kc, tau = [], 0.0, 0.0
    try:
        for it in range(n_iter):
            batch = loader.next_batch(bs, 'valid', it * bs)
            c = cgn_model.discriminator_forward(batch, bs, use_cuda)
            ce, kld, z_loss, c_loss, kc, tau, enc_loss, gen_loss = valid_enc_gen(batch, bs, use_cuda, c_target=c,
                                                                                 global_step=step, calc_z_loss=False)
            rows.append((enc_loss, gen_loss, z_loss, c_loss, ce, kld))
    finally:
        cgn_model.prep_enc_gen_training()  # :222 back to training mode
    if not rows:
        return {}
    means = _mean_on_device(rows)
    names = ("encoder_loss", "generator_loss", "generator_z_loss", "generator_c_loss", "cross_entropy", "kld")
    return _Scalars(summary_writer, verbose).emit("valid_enc_gen", step,
                                                  list(zip(names, means)) + [("kld_coef", kc), ("temp_coef", tau)])


def disc_validation(cgn_model, data_handler, config, use_cuda, step, num_valid_batches, sentiment_disc_eval_step,
                    number_of_gen_samples, summary_writer, verbose=True):
    """train_functions.py:224-263: labelled dev batches + freshly sampled sentences through the discriminator; mean
    losses and the two accuracies in percent (:245-246); returns {name: value}."""
    cgn_model.prep_disc_validation()
    rows = []
    try:
        for b in range(num_valid_batches):
            samples, codes = generate_samples_for_disc_training(cgn_model, data_handler, config, use_cuda,
                                                                number_of_gen_samples, GEN_SAMPLES_PER_CALL)
            ce_l, ce_g, n_ok, n_ok_gen = sentiment_disc_eval_step(b, samples, codes)
            rows.append((ce_l, ce_g, n_ok.float(), n_ok_gen.float()))
    finally:
        cgn_model.prep_disc_training()  # :263
    if not rows:
        return {}
    ce_l, ce_g, ok, ok_gen = _mean_on_device(rows)
    C = getattr(config, "c_size", 1)  # every attribute's prediction is counted when c has several columns
    pairs = [("labeled_cross_entropy", ce_l), ("labeled_accuracy", ok * (100.0 / (data_handler.batch_size * C))),
             ("generated_cross_entropy", ce_g), ("generated_accuracy", ok_gen * (100.0 / (number_of_gen_samples * C)))]
    return _Scalars(summary_writer, verbose).emit("valid_discriminator", step, pairs)


def train_complete_model(cgn_model, data_handler, config, use_cuda, dropout, summary_writer, save_model_dir, model_name,
                         num_steps=2500000, log_every=0, validate=None):
    """train_functions.py:19-108. One iteration = sample B sentences -> discriminator step -> c inference ->
    encoder/generator step (calc_z_loss=False, :90). Returns the last iteration's (discriminator 5-tuple,
    encoder/generator 8-tuple) as device scalars.
    log_every: 0 = quiet (no host syncs between iterations); n > 0 = print/log losses every n-th iteration instead of
    the reference's fixed 5 / 50. validate: None = run the reference's validation schedule (every 40th / 100th
    iteration) only when a summary writer or log_every asks for output; True / False force it."""
    for switch in (cgn_model.discriminator_mode, cgn_model.encoder_mode, cgn_model.generator_mode):
        switch(train_mode=True)  # :24-26
    disc_step = cgn_model.discriminator_sentiment_trainer(data_handler, use_cuda)
    disc_eval = cgn_model.discriminator_sentiment_valid(data_handler, use_cuda, return_accuracy=True)
    enc_gen_step = cgn_model.train_encoder_generator(summary_writer)
    enc_gen_eval = cgn_model.valid_encoder_generator()

    B = data_handler.batch_size
    n_disc = data_handler.num_train_sentiment_batches
    n_gen = (data_handler.gen_batch_loader.train_lines // B)  # whole batches only (:42-44)
    if n_disc < 1 or n_gen < 1:
        raise ValueError("need at least one labelled batch and one generator batch of %d sentences" % B)
    talk = bool(log_every) or summary_writer is not None
    validate = talk if validate is None else bool(validate)
    out = _Scalars(summary_writer, verbose=bool(log_every))
    d_every = log_every or DISC_LOG_EVERY
    g_every = log_every or ENC_GEN_LOG_EVERY
    last = None
    for step in range(num_steps):
        # ---- wake: discriminator on labelled + generated sentences (:55-57)
        cgn_model.prep_disc_training()
        samples, codes = generate_samples_for_disc_training(cgn_model, data_handler, config, use_cuda, B,
                                                            GEN_SAMPLES_PER_CALL)
        d_losses = disc_step(samples, codes, step % n_disc)
        if talk and step % d_every == 0:
            out.emit("train_discriminator", step, zip(_DISC_TAGS, d_losses))
        if validate and step % DISC_VALID_EVERY == 0:
            disc_validation(cgn_model, data_handler, config, use_cuda, step, data_handler.num_dev_sentiment_batches,
                            disc_eval, B, summary_writer, verbose=bool(log_every))

        # ---- sleep: encoder/generator on a corpus batch with c inferred by the (frozen) discriminator (:77-90)
        batch = data_handler.gen_batch_loader.next_batch(B, 'train', B * (step % n_gen))
        cgn_model.discriminator_mode(train_mode=False)
        c = cgn_model.discriminator_forward(batch, B, use_cuda)
        cgn_model.prep_enc_gen_training()
        eg = enc_gen_step(batch, B, use_cuda, dropout, c, step, calc_z_loss=False)
        if talk and step % g_every == 0:
            ce, kld, z_loss, c_loss, kc, tau, enc_loss, gen_loss = eg
            out.emit("train_enc_gen", step, [("encoder_loss", enc_loss), ("generator_loss", gen_loss),
                                            ("generator_z_loss", z_loss), ("generator_c_loss", c_loss),
                                            ("cross_entropy", ce), ("kld", kld), ("kld_coeff", kc), ("temp_coef", tau)])
        if validate and step % ENC_GEN_VALID_EVERY == 0:
            enc_gen_validation(cgn_model, data_handler, use_cuda, step, enc_gen_eval, summary_writer,
                               verbose=bool(log_every))
        last = (d_losses, eg)

        if save_model_dir is not None and (step + 1) % CHECKPOINT_EVERY == 0:
            _ops.raise_on_lstm_timeout()  # never write weights that were trained on a timed-out launch
            t.save(cgn_model.state_dict(), os.path.join(save_model_dir, 'cgn_model_experiment_' + model_name))
    if last is not None:
        _ops.raise_on_lstm_timeout()
    return last
