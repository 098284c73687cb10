"""The conditioning side of a training iteration, kept behind the reference's own call contract.

Reference: ``LatentDiffusion.shared_step`` ddpm.py:1436-1938 (iteration flags, which of the batch's prompt lists are used,
zero-shot features), ``LatentDiffusion.forward`` ddpm.py:1940-2179 (four-way prompt list -> ``get_learned_conditioning`` ->
chunk(4), class-prompt patching, ``c_static_emb`` selection, ``extra_info`` assembly) and ``get_learned_conditioning``
ddpm.py:970-1085 (the call into ``FrozenCLIPEmbedder.encode(cond_in, embedding_manager=...)`` -> ``EmbeddingManager.forward`` ->
``SubjBasisGenerator.forward``).  Those three callees stay the reference's own classes -- instantiated from the yaml's
``cond_stage_config`` / ``personalization_config`` through ``instantiate_from_config`` -- and are only CALLED here, with the
arguments, in the order and with the embedding-manager state updates the reference makes.  Host-side bookkeeping only: no
kernel is launched from this file; what it produces is the ``cond = (c_static_emb, c_in, extra_info)`` triple the UNet consumes.
"""
import copy
import random

import numpy as np
import torch
import torch.nn.functional as F

from ...util import (anneal_add_noise_to_embedding, anneal_value, distribute_embedding_to_M_tokens_by_dict, halve_token_indices,
                     join_dict_of_indices_with_key_filter, merge_cls_token_embeddings, repeat_selected_instances)

# the prompt lists a batch carries (ldm/data/personalized.py:870-981), by (use_fp_trick, use_background_token)
_PROMPT_KEYS = {
    (True, True): ("caption_bg", "subj_prompt_single_fp_bg", "subj_prompt_comp_fp_bg", "cls_prompt_single_fp_bg", "cls_prompt_comp_fp_bg"),
    (True, False): ("caption", "subj_prompt_single_fp", "subj_prompt_comp_fp", "cls_prompt_single_fp", "cls_prompt_comp_fp"),
    (False, True): ("caption_bg", "subj_prompt_single_bg", "subj_prompt_comp_bg", "cls_prompt_single_bg", "cls_prompt_comp_bg"),
    (False, False): ("caption", "subj_prompt_single", "subj_prompt_comp", "cls_prompt_single", "cls_prompt_comp"),
}


class ConditioningMixin:
    """methods of ``LatentDiffusion`` that talk to ``self.cond_stage_model`` / ``self.embedding_manager``."""

    N_CA_LAYERS = 16

    # ---- ddpm.py:970-1085 ------------------------------------------------------------------------------------------
    def get_learned_conditioning(self, cond_in, zs_clip_features=None, zs_id_embs=None, zs_out_id_embs_scale_range=(1.0, 1.0),
                                 randomize_clip_weights=False, apply_arc2face_inverse_embs=False,
                                 apply_arc2face_embs=False, embman_iter_type=None):
        """prompts -> ``(static_prompt_embedding [16*len(cond_in), 77, 768], cond_in, extra_info)``.  Same signature, same
        sequence of calls into the text encoder and the embedding manager, same ``extra_info`` keys as the reference."""
        if getattr(self, "cond_stage_forward", None) is not None:
            return getattr(self.cond_stage_model, self.cond_stage_forward)(cond_in)
        csm, em = self.cond_stage_model, self.embedding_manager
        if not (hasattr(csm, "encode") and callable(csm.encode)):
            return csm(cond_in)
        dev = self.device
        csm.device = dev
        if self.empty_context is not None:
            self.empty_context = self.empty_context.to(dev)
        if randomize_clip_weights:
            csm.sample_last_layers_skip_weights()
        if zs_clip_features is not None or zs_id_embs is not None:
            em.set_zs_image_features(zs_clip_features, zs_id_embs, zs_out_id_embs_scale_range=zs_out_id_embs_scale_range,
                                     add_noise_to_zs_id_embs=not self.iter_flags["add_noise_to_real_id_embs"])
            apply_compel_cfg_prob = 0           # zero-shot: compel cfg was applied inside the embedding manager
        else:
            apply_compel_cfg_prob = self.apply_compel_cfg_prob
        if embman_iter_type is None:
            if self.iter_flags["is_compos_iter"]:
                embman_iter_type = "compos_distill_iter"
            elif apply_arc2face_inverse_embs:
                embman_iter_type = "arc2face_inverse_clip_iter"
            elif apply_arc2face_embs:
                embman_iter_type = "arc2face_clip_iter"
            else:
                embman_iter_type = "recon_iter"
        em.set_curr_iter_type(embman_iter_type)
        static_prompt_embedding = csm.encode(cond_in, embedding_manager=em)           # -> EmbeddingManager -> SBG hook
        if apply_arc2face_embs:
            static_prompt_embedding = em.arc2face_embs
        if self.training and not apply_arc2face_inverse_embs:
            static_prompt_embedding = merge_cls_token_embeddings(static_prompt_embedding, em.cls_delta_string_indices,
                                                                 em.subj_name_to_cls_delta_token_weights)
        elif apply_arc2face_inverse_embs or apply_arc2face_embs:
            reps = 1 if not self.training else len(cond_in) // static_prompt_embedding.shape[0]
            static_prompt_embedding = static_prompt_embedding.unsqueeze(1).repeat(reps, 16, 1, 1) \
                .reshape(-1, *static_prompt_embedding.shape[1:])
        extra_info = {"use_layerwise_context": self.use_layerwise_embedding,
                      "use_conv_attn_kernel_size": em.use_conv_attn_kernel_size,
                      "placeholder2indices": copy.copy(em.placeholder2indices),
                      "prompt_emb_mask": copy.copy(em.prompt_emb_mask),
                      "is_training": em.training,
                      "compel_cfg_weight_level_range": self.compel_cfg_weight_level_range,
                      "apply_compel_cfg_prob": apply_compel_cfg_prob,
                      "empty_context": self.empty_context,
                      "capture_distill_attn": False}
        return static_prompt_embedding, cond_in, extra_info

    # ---- ddpm.py:1940-2179: everything of ``forward`` before ``p_losses`` ---------------------------------------------
    def assemble_conditioning(self, captions, orig_bs):
        """-> ``cond = (c_static_emb, c_in, extra_info)`` for ``orig_bs`` instances, from ``self.iter_flags`` as the front of
        ``shared_step`` left them (``delta_prompts`` = four lists, ``zs_clip_features``, ``zs_id_embs``, the iteration type)."""
        fl = self.iter_flags
        inverse = fl["do_arc2face_distill"] and self.apply_arc2face_inverse_embs
        if not (fl["do_static_prompt_delta_reg"] or fl["do_mix_prompt_distillation"]):
            # recon iteration without the static delta loss (ablation / validation): subject-single prompts only
            assert fl["do_normal_recon"]
            c_static_emb, c_in, extra_info = self.get_learned_conditioning(
                captions, fl["zs_clip_features"], fl["zs_id_embs"], randomize_clip_weights=True,
                apply_arc2face_inverse_embs=inverse)
            extra_info["placeholder2indices_1b"] = extra_info["placeholder2indices"]
            extra_info["c_static_emb_1b"] = c_static_emb.reshape(orig_bs, self.N_CA_LAYERS, *c_static_emb.shape[1:])
            extra_info["iter_type"] = "normal_recon"
            return c_static_emb, c_in, extra_info

        if fl.get("reuse_init_conds"):
            delta_prompts = self.cached_inits[self.batch_1st_subject_name]["delta_prompts"]
        else:
            delta_prompts = fl["delta_prompts"]
        subj_single, subj_comp, cls_single, cls_comp = delta_prompts
        if fl["do_mix_prompt_distillation"] or fl["do_ada_prompt_delta_reg"]:
            block = 1                                                   # compositional iterations work on ONE instance
            subj_single, subj_comp, cls_single, cls_comp = (p[:block] for p in (subj_single, subj_comp, cls_single, cls_comp))
        else:
            block = orig_bs
        four_way = list(subj_single) + list(subj_comp) + list(cls_single) + list(cls_comp)
        c_static_emb, _, extra_info = self.get_learned_conditioning(
            four_way, fl["zs_clip_features"], fl["zs_id_embs"], randomize_clip_weights=True, apply_arc2face_inverse_embs=inverse)
        subj_single_emb, subj_comp_emb, cls_single_emb, cls_comp_emb = c_static_emb.chunk(4)
        # the placeholder appears only in the two subject blocks; halving leaves the subject-single block's indices,
        # which also address the (aligned) class prompts
        ph2 = extra_info["placeholder2indices"]
        ph1 = {k: halve_token_indices(v) for k, v in ph2.items()}
        # the class prompts hold "person , , ," where the subject prompts hold the K subject embeddings: spread the class
        # token's embedding over those K positions
        cls_single_emb = distribute_embedding_to_M_tokens_by_dict(cls_single_emb, ph1)
        cls_comp_emb = distribute_embedding_to_M_tokens_by_dict(cls_comp_emb, ph1)
        extra_info["placeholder2indices_1b"] = ph1
        extra_info["placeholder2indices_2b"] = ph2
        c_static_emb = torch.cat([subj_single_emb, subj_comp_emb, cls_single_emb, cls_comp_emb], dim=0)
        extra_info["c_static_emb_4b"] = c_static_emb.reshape(4 * block, self.N_CA_LAYERS, *c_static_emb.shape[1:])
        if fl["do_mix_prompt_distillation"]:
            c_in2 = four_way
            extra_info["iter_type"] = self.prompt_mix_scheme                        # 'mix_hijk'
            extra_info["placeholder2indices"] = ph2
        elif fl["do_ada_prompt_delta_reg"]:
            c_in2 = four_way
            extra_info["iter_type"] = "do_ada_prompt_delta_reg"
            extra_info["placeholder2indices"] = ph2
        else:
            extra_info["iter_type"] = "normal_recon"
            c_in2 = captions
            if list(captions) != list(subj_single):
                raise RuntimeError("recon iteration: captions differ from the subject-single prompts (the reference stops "
                                   "here too, ddpm.py:2113)")
            c_static_emb = subj_single_emb
            extra_info["placeholder2indices"] = ph1
            extra_info["c_static_emb_1b"] = c_static_emb.reshape(orig_bs, self.N_CA_LAYERS, *c_static_emb.shape[1:])
        extra_info["cls_single_prompts"] = cls_single
        extra_info["cls_comp_prompts"] = cls_comp
        extra_info["delta_prompts"] = (subj_single, subj_comp, cls_single, cls_comp)
        return c_static_emb, c_in2, extra_info

    def attach_subject_indices(self, extra_info):
        """ddpm.py:2566-2573: the (instance, token) positions of the subject / background embeddings =
        ``placeholder2indices`` joined over the embedding manager's subject / background strings -- what the recon
        iteration's attention losses read (``recon_regularizers``)."""
        em = self.embedding_manager
        extra_info["subj_indices"] = join_dict_of_indices_with_key_filter(extra_info["placeholder2indices"], em.subject_string_dict)
        extra_info["bg_indices"] = join_dict_of_indices_with_key_filter(extra_info["placeholder2indices"], em.background_string_dict)
        extra_info["subj_indices_1b"] = join_dict_of_indices_with_key_filter(extra_info.get("placeholder2indices_1b"),
                                                                             em.subject_string_dict)
        extra_info["bg_indices_1b"] = join_dict_of_indices_with_key_filter(extra_info.get("placeholder2indices_1b"),
                                                                           em.background_string_dict)
        if "placeholder2indices_2b" in extra_info:
            extra_info["subj_indices_2b"] = join_dict_of_indices_with_key_filter(extra_info["placeholder2indices_2b"],
                                                                                 em.subject_string_dict)
            extra_info["bg_indices_2b"] = join_dict_of_indices_with_key_filter(extra_info["placeholder2indices_2b"],
                                                                               em.background_string_dict)
        return extra_info

    # ---- ddpm.py:1436-1938: the front of ``shared_step`` for recon / Arc2Face-distillation iterations -------------------
    def prepare_recon_iteration(self, batch, x_start, img_mask, fg_mask, py_random=random, np_random=np.random):
        """Sets the iteration flags, picks the prompt lists, obtains the zero-shot features and (distillation) the teacher's
        prompt embedding, trims a multi-step distillation batch to HALF_BS -- consuming ``random`` / ``np.random`` in the
        reference's order (one draw each for: use_wds_comp, use_background_token, gen_arc2face_rand_face,
        add_noise_to_real_id_embs, use_arc2face_as_target, num_denoising_steps, as far as their branch is reached).
        -> (x_start, img_mask, fg_mask, captions).  wds-overlay iterations and compositional iterations are not handled here."""
        fl = self.iter_flags
        self.embedding_manager.training_percent = self.training_percent
        have = batch["has_fg_mask"]
        fl["fg_mask_avail_ratio"] = have.sum() / have.shape[0]
        wds = batch.get("has_wds_comp")
        fl["wds_comp_avail_ratio"] = 0 if wds is None else wds.sum() / wds.shape[0]
        self.batch_1st_subject_name = batch["subject_name"][0]
        fl["reuse_init_conds"] = False
        fl["do_teacher_filter"] = False
        captions = batch["caption"]
        delta_prompts = None
        if self.do_static_prompt_delta_reg or fl["do_mix_prompt_distillation"]:
            fl["use_fp_trick"] = False                                    # only compositional iterations use it
            if fl["wds_comp_avail_ratio"] == 1:
                raise NotImplementedError("wds-overlay batches are out of scope (SURVEY.md section 2: data pipeline)")
            fl["use_wds_comp"] = py_random.random() < 0                   # p_use_wds_comp = 0; the draw is still made
            fl["comp_init_fg_from_training_image"] = False
            p_bg = 0 if fl["do_arc2face_distill"] else 0.9
            fl["use_background_token"] = bool(self.use_background_token and py_random.random() < p_bg)
            cap_key, k_ss, k_sc, k_cs, k_cc = _PROMPT_KEYS[(False, fl["use_background_token"])]
            captions = batch[cap_key]
            subj_single, cls_single = list(batch[k_ss]), list(batch[k_cs])
            subj_comp = [p.split("|")[0] for p in batch[k_sc]]
            cls_comp = [p.split("|")[0] for p in batch[k_cc]]
            delta_prompts = (subj_single, subj_comp, cls_single, cls_comp)
        elif fl["use_background_token"]:
            captions = batch["caption_bg"]
        BS = len(batch["subject_name"])
        fl["same_subject_in_batch"] = False
        if fl["do_arc2face_distill"] and py_random.random() < self.p_gen_arc2face_rand_face:
            fl["gen_arc2face_rand_face"] = True
            self.batch_subject_names = ["arc2face"] * BS
        else:
            fl["gen_arc2face_rand_face"] = False
            self.batch_subject_names = list(batch["subject_name"])
        being_faces = self.embedding_manager.subj_name_to_being_faces
        fl["is_face"] = [being_faces[n] for n in self.batch_subject_names]
        arc2face_prompt_emb = None
        zs_clip_features = zs_id_embs = None
        if self.do_zero_shot:
            if not fl["gen_arc2face_rand_face"]:
                zs_clip_features, zs_id_embs, faceless = self.zero_shot_features(batch, fg_mask, is_face=fl["is_face"][0],
                                                                                 calc_avg=fl["same_subject_in_batch"])
                p_noise = self.p_add_noise_to_real_id_embs if fl["do_arc2face_distill"] else 0
                fl["add_noise_to_real_id_embs"] = py_random.random() < p_noise
                if fl["add_noise_to_real_id_embs"]:
                    fl["same_subject_in_batch"] = True
                    (x_start, img_mask, fg_mask, have, self.batch_subject_names, fl["is_face"], zs_clip_features,
                     zs_id_embs) = repeat_selected_instances(slice(0, 1), BS, x_start, img_mask, fg_mask, have,
                                                             self.batch_subject_names, fl["is_face"], zs_clip_features,
                                                             zs_id_embs)
                    zs_id_embs = anneal_add_noise_to_embedding(zs_id_embs, 0, begin_noise_std_range=[0.02, 0.06],
                                                               end_noise_std_range=None, add_noise_prob=1,
                                                               noise_std_is_relative=True, keep_norm=True)
                fl["faceless_img_count"] = faceless
                if fl["do_arc2face_distill"]:
                    _, _, arc2face_prompt_emb = self.arc2face.gen_arc2face_prompt_embs(x_start.shape[0], pre_face_embs=zs_id_embs)
            else:
                zs_clip_features = torch.zeros(x_start.shape[0], 514, 1280, device=x_start.device)
                _, zs_id_embs, arc2face_prompt_emb = self.arc2face.gen_arc2face_prompt_embs(x_start.shape[0], pre_face_embs=None)
                img_mask = fg_mask = None
                have = torch.zeros_like(have)
                x_start = torch.randn_like(x_start)
                fl["is_face"] = [True] * x_start.shape[0]
                fl["faceless_img_count"] = 0
            zs_id_embs = zs_id_embs.to(x_start.dtype)
            if fl["do_arc2face_distill"]:
                arc2face_prompt_emb = arc2face_prompt_emb.to(x_start.dtype)
                if fl["gen_arc2face_rand_face"] or fl["add_noise_to_real_id_embs"] or fl["faceless_img_count"] > 0:
                    fl["use_arc2face_as_target"] = True
                else:
                    fl["use_arc2face_as_target"] = py_random.random() < 0.5
                if fl["use_arc2face_as_target"]:
                    nd = self.draw_num_denoising_steps(self.max_num_denoising_steps, np_random)
                    fl["num_denoising_steps"] = nd
                    if nd > 1:
                        hb = self.half_batch_size(BS, nd)
                        (x_start, img_mask, fg_mask, have, self.batch_subject_names, captions, fl["is_face"],
                         zs_clip_features, zs_id_embs, arc2face_prompt_emb) = repeat_selected_instances(
                            slice(0, hb), 1, x_start, img_mask, fg_mask, have, self.batch_subject_names, captions,
                            fl["is_face"], zs_clip_features, zs_id_embs, arc2face_prompt_emb)
                        if delta_prompts is not None:
                            delta_prompts = tuple(p[:hb] for p in delta_prompts)
            else:
                fl["use_arc2face_as_target"] = False
        fl.update(img_mask=img_mask, fg_mask=fg_mask, batch_have_fg_mask=have, delta_prompts=delta_prompts,
                  zs_clip_features=zs_clip_features, zs_id_embs=zs_id_embs, arc2face_prompt_emb=arc2face_prompt_emb)
        embman_iter_type = "arc2face_inverse_clip_iter" if (fl["do_arc2face_distill"] and self.apply_arc2face_inverse_embs) \
            else "recon_iter"
        self.embedding_manager.set_curr_batch_subject_names(self.batch_subject_names, embman_iter_type)
        return x_start, img_mask, fg_mask, captions

    def prepare_compos_iteration(self, batch, x_start, img_mask, fg_mask, py_random=random):
        """the front of ``shared_step`` for a compositional (prompt-mix distillation) iteration, ddpm.py:1450-1936: whether a
        cached first-pass result of this subject is reused, the face-portrait prompt trick, the foreground-initialised
        latent, the background token -- ``random`` consumed in the reference's order -- then the whole batch becomes BS
        copies of its FIRST instance (one subject per compositional iteration) and the zero-shot features are averaged.
        -> (x_start, img_mask, fg_mask, captions)."""
        fl = self.iter_flags
        assert fl["is_compos_iter"] and fl["do_mix_prompt_distillation"]
        self.embedding_manager.training_percent = self.training_percent
        have = batch["has_fg_mask"]
        fl["fg_mask_avail_ratio"] = have.sum() / have.shape[0]
        wds = batch.get("has_wds_comp")
        fl["wds_comp_avail_ratio"] = 0 if wds is None else wds.sum() / wds.shape[0]
        self.batch_1st_subject_name = batch["subject_name"][0]
        in_mix_folder = bool(batch.get("is_in_mix_subj_folder", [False])[0])
        p_reuse = 0.25 if in_mix_folder else 1
        fl["reuse_init_conds"] = bool(self.batch_1st_subject_name in self.cached_inits and py_random.random() < p_reuse)
        fl["do_teacher_filter"] = bool(self.do_clip_teacher_filtering and not fl["reuse_init_conds"])
        fl["use_fp_trick"] = bool(self.use_fp_trick and "subj_prompt_single_fp" in batch and py_random.random() < 0.9)
        if fl["wds_comp_avail_ratio"] == 1:
            raise NotImplementedError("wds-overlay batches are out of scope (SURVEY.md section 2: data pipeline)")
        fl["use_wds_comp"] = py_random.random() < 0
        p_init_fg = 1 if self.do_zero_shot else anneal_value(self.training_percent, 0.5, (0.7, 0.9))
        fl["comp_init_fg_from_training_image"] = bool(not fl["reuse_init_conds"] and fl["fg_mask_avail_ratio"] > 0
                                                      and py_random.random() < p_init_fg)
        fl["use_background_token"] = bool(self.use_background_token and py_random.random() < 0.5)
        cap_key, k_ss, k_sc, k_cs, k_cc = _PROMPT_KEYS[(fl["use_fp_trick"], fl["use_background_token"])]
        captions = batch[cap_key]
        delta_prompts = (list(batch[k_ss]), [p.split("|")[0] for p in batch[k_sc]], list(batch[k_cs]),
                         [p.split("|")[0] for p in batch[k_cc]])
        BS = len(batch["subject_name"])
        names, x_start, img_mask, fg_mask, have = repeat_selected_instances(
            slice(0, 1), BS, list(batch["subject_name"]), x_start, img_mask, fg_mask, have)
        fl["same_subject_in_batch"] = True
        fl["gen_arc2face_rand_face"] = False
        self.batch_subject_names = names
        fl["is_face"] = [self.embedding_manager.subj_name_to_being_faces[n] for n in names]
        zs_clip_features = zs_id_embs = None
        if self.do_zero_shot:
            first = {k: (v[:1].repeat(BS, *([1] * (v.dim() - 1))) if torch.is_tensor(v) and v.dim() > 0 and v.shape[0] == BS else v)
                     for k, v in batch.items()}
            zs_clip_features, zs_id_embs, faceless = self.zero_shot_features(first, fg_mask, is_face=fl["is_face"][0], calc_avg=True)
            fl["add_noise_to_real_id_embs"] = py_random.random() < 0          # (the draw is made; p = 0 outside distillation)
            fl["faceless_img_count"] = faceless
            zs_id_embs = zs_id_embs.to(x_start.dtype)
            fl["use_arc2face_as_target"] = False
        fl.update(img_mask=img_mask, fg_mask=fg_mask, batch_have_fg_mask=have, delta_prompts=delta_prompts,
                  zs_clip_features=zs_clip_features, zs_id_embs=zs_id_embs, arc2face_prompt_emb=None)
        if fl["reuse_init_conds"]:
            cached = self.cached_inits[self.batch_1st_subject_name]
            for k in ("delta_prompts", "img_mask", "fg_mask", "batch_have_fg_mask", "filtered_fg_mask", "use_background_token",
                      "use_wds_comp", "comp_init_fg_from_training_image", "zs_clip_features", "zs_id_embs", "arc2face_prompt_emb"):
                fl[k] = cached[k]
        self.embedding_manager.set_curr_batch_subject_names(self.batch_subject_names, "compos_distill_iter")
        return x_start, img_mask, fg_mask, captions

    def zero_shot_features(self, batch, fg_mask, is_face=True, calc_avg=False):
        """``encode_zero_shot_image_features`` (ddpm.py:2322-2471) is CLIP-vision + insightface / DINO inference on
        third-party weights -- a boundary callee.  A batch that already carries ``zs_clip_features`` [B,514,D] and
        ``zs_id_embs`` [B,512] (the synthetic batches of the bench do) is used as is; otherwise the attached
        ``zero_shot_encoder(images_u8_nchw, fg_mask, is_face=, calc_avg=)`` callable is asked."""
        if "zs_clip_features" in batch and "zs_id_embs" in batch:
            f, e = batch["zs_clip_features"], batch["zs_id_embs"]
            if calc_avg:                 # ddpm.py:2442-2465: mean over the instances, the id embedding re-normalised
                f = f.mean(dim=0, keepdim=True)
                e = torch.nn.functional.normalize(e.mean(dim=0, keepdim=True), p=2, dim=-1)
            return f, e, 0
        images = batch["image_unnorm"].permute(0, 3, 1, 2)
        masks = None if fg_mask is None else fg_mask.squeeze(1)
        if getattr(self, "clip_image_encoder", None) is not None:                 # the reference's own front end, mirrored
            return self.encode_zero_shot_image_features(images, masks, image_paths=batch.get("image_path"), is_face=is_face,
                                                        calc_avg=calc_avg)
        enc = getattr(self, "zero_shot_encoder", None)
        if enc is None:
            raise RuntimeError("the batch carries no zs_clip_features / zs_id_embs and no image encoders are attached "
                               "(set_zero_shot_image_encoders) nor a zero_shot_encoder callable")
        return enc(images, masks, is_face=is_face, calc_avg=calc_avg)

    # ---- the zero-shot feature front end (ddpm.py:904-941, 2322-2471) -----------------------------------
    def set_zero_shot_image_encoders(self, clip_image_encoder, clip_preprocessor, insightface_app=None, dino_encoder=None,
                                     dino_preprocess=None):
        """What ``instantiate_zero_shot_image_encoders`` (ddpm.py:904-941) builds from downloaded third-party weights, handed
        over instead: ``clip_image_encoder(pixel_values, attn_mask=, output_hidden_states=True)`` with the reference's
        ``CLIPVisionModelWithMask`` contract -- ``adaprompt_amd.clip_vision.CLIPVisionModelWithMask`` is that model on the
        MI355X kernels; ``clip_preprocessor(images=, return_tensors="pt").pixel_values`` (HF ``CLIPImageProcessor``);
        ``insightface_app.get(bgr_image)`` (ArcFace, ONNX) and the DINO pair for non-face subjects stay third-party."""
        self.clip_image_encoder, self.clip_preprocessor = clip_image_encoder, clip_preprocessor
        self.insightface_app, self.dino_encoder, self.dino_preprocess = insightface_app, dino_encoder, dino_preprocess
        self.neg_image_features = None
        self.zs_image_encoders_instantiated = True

    def instantiate_zero_shot_image_encoders(self, clip_type="openai"):
        raise RuntimeError("the zero-shot image encoders need third-party checkpoints (HF CLIP, insightface antelopev2, DINO): "
                           "build them and hand them over with set_zero_shot_image_encoders(...)")

    def encode_zero_shot_image_features(self, images, fg_masks, image_paths=None, is_face=True, size=(512, 512), calc_avg=False,
                                        skip_non_faces=False, verbose=False):
        """ddpm.py:2322-2471.  Per image: CLIP pixel values from the preprocessor; the identity embedding from insightface
        (the largest detected face; a random vector when none is found, or the image is skipped) or DINO's class token for
        non-faces.  Then the image encoder runs twice under no_grad -- foreground mask, background mask (1 - mask) -- and once,
        cached, on an all-zero image; the features are the penultimate hidden states minus the zero image's, times the token
        mask, concatenated along tokens: [BS, 514, D].  ``calc_avg``: mean over the batch, identity embedding re-normalised.
        -> (clip_features, id_embs, faceless_img_count)."""
        if not getattr(self, "zs_image_encoders_instantiated", False):
            self.instantiate_zero_shot_image_encoders()
        dev = self.device
        pixel_values, all_id_embs, faceless = [], [], 0
        for idx, image in enumerate(images):
            pixel_values.append(self.clip_preprocessor(images=image, return_tensors="pt").pixel_values)
            if is_face and self.insightface_app is not None:
                if isinstance(image, torch.Tensor):
                    image = image.cpu().numpy().transpose(1, 2, 0)
                from PIL import Image
                image = np.array(Image.fromarray(image).resize(size, Image.NEAREST))
                faces = self.insightface_app.get(np.ascontiguousarray(image[..., ::-1]))           # RGB -> BGR
                if len(faces) == 0 and not skip_non_faces:
                    print(f"No face detected in {image_paths[idx]}. Use random face embedding.")
                    id_emb = torch.randn(512, device=dev)
                    faceless += 1
                elif len(faces) > 0:
                    # (the reference's key is (x2 - x1) * y2 - y1 -- precedence as written there, ddpm.py:2355)
                    face = sorted(faces, key=lambda f: (f["bbox"][2] - f["bbox"][0]) * f["bbox"][3] - f["bbox"][1])[-1]
                    id_emb = torch.from_numpy(face.normed_embedding).to(dev)
                else:
                    print(f"Skip image without face: {image_paths[idx]}")
                    continue
                all_id_embs.append(id_emb)
            elif not is_face:
                dino_input = self.dino_preprocess(images=image, return_tensors="pt").to(dev)
                all_id_embs.append(self.dino_encoder(**dino_input).last_hidden_state[0, 0])
        if verbose:
            print(f"{len(all_id_embs)} face images identified, {faceless} faceless images.")
        pixel_values = torch.cat(pixel_values, dim=0).to(dev)
        all_id_embs = torch.stack(all_id_embs, dim=0) if self.insightface_app is not None else None
        hw = pixel_values.shape[-2:]
        if fg_masks is not None:
            assert len(fg_masks) == len(images)
            if isinstance(fg_masks, (list, tuple)):
                m2 = [F.interpolate(torch.as_tensor(m, device=dev).float()[None, None], size=hw, mode="bilinear", align_corners=False)
                      for m in fg_masks]
                fg_masks2 = torch.cat(m2, dim=0).squeeze(1)
            else:
                fg_masks2 = F.interpolate(torch.as_tensor(fg_masks, device=dev).float().unsqueeze(1), size=hw, mode="bilinear",
                                          align_corners=False).squeeze(1)
        else:
            fg_masks2 = torch.ones_like(pixel_values[:, 0])
        enc = self.clip_image_encoder
        # only the penultimate hidden states are read: an encoder that can stop there is asked to (the MI355X one can)
        n_layers = getattr(enc, "stop_before_last_layer", None)
        kw, pen = ({}, -2) if n_layers is None else ({"layers_needed": n_layers}, -1)
        with torch.no_grad():
            if self.neg_image_features is None:
                self.neg_image_features = enc(torch.zeros_like(pixel_values[:1]).half(), attn_mask=None,
                                              output_hidden_states=True, **kw).hidden_states[pen]
            feats = []
            if n_layers is not None:
                # the MI355X encoder: foreground and background pass as ONE batch of 2 BS (samples are independent; twice
                # the rows per contraction, half the launches)
                pv2 = torch.cat([pixel_values.half(), pixel_values.half()], dim=0)
                out = enc(pv2, attn_mask=torch.cat([fg_masks2.half(), 1 - fg_masks2.half()], dim=0), output_hidden_states=True, **kw)
                f = (out.hidden_states[pen] - self.neg_image_features) * out.attn_mask
                feats = list(f.chunk(2, dim=0))
            else:
                for mask in (fg_masks2.half(), 1 - fg_masks2.half()):
                    out = enc(pixel_values.half(), attn_mask=mask, output_hidden_states=True, **kw)
                    f = out.hidden_states[pen] - self.neg_image_features
                    if out.attn_mask is not None:
                        f = f * out.attn_mask
                    feats.append(f)
        clip_features = torch.cat(feats, dim=1).to(pixel_values.dtype)
        if calc_avg:
            clip_features = clip_features.mean(dim=0, keepdim=True)
            id_embs = None if all_id_embs is None else F.normalize(all_id_embs.mean(dim=0, keepdim=True), p=2, dim=-1)
        else:
            id_embs = all_id_embs
        return clip_features, id_embs, faceless
